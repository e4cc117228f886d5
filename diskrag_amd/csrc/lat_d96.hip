#define DR_DIM 96
#define DR_PART_LAT 1
#include "search_dim.inc"
