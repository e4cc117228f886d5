#define DR_DIM 256
#include "search_dim.inc"
