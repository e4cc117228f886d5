#define DR_DIM 960
#include "search_dim.inc"
