#define DR_DIM 32
#include "search_dim.inc"
