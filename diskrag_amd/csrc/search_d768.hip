#define DR_DIM 768
#include "search_dim.inc"
