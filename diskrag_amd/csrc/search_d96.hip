#define DR_DIM 96
#include "search_dim.inc"
