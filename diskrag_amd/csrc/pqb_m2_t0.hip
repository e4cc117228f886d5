#define PQB_M16 2
#define PQB_TREG 0
#include "pqb_tu.inc"
