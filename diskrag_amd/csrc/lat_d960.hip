#define DR_DIM 960
#define DR_PART_LAT 1
#include "search_dim.inc"
