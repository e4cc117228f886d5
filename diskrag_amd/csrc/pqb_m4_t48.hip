#define PQB_M16 4
#define PQB_TREG 48
#include "pqb_tu.inc"
