#define PQB_M16 2
#define PQB_TREG 16
#include "pqb_tu.inc"
