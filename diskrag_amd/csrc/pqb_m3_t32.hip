#define PQB_M16 3
#define PQB_TREG 32
#include "pqb_tu.inc"
