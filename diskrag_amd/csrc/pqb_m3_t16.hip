#define PQB_M16 3
#define PQB_TREG 16
#include "pqb_tu.inc"
