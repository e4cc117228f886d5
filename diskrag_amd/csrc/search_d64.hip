#define DR_DIM 64
#include "search_dim.inc"
