#define DR_DIM 128
#include "search_dim.inc"
