#define PQB_M16 4
#define PQB_TREG 32
#include "pqb_tu.inc"
