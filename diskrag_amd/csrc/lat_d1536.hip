#define DR_DIM 1536
#define DR_PART_LAT 1
#include "search_dim.inc"
