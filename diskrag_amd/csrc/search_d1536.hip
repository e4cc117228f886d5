#define DR_DIM 1536
#include "search_dim.inc"
