#define PQB_M16 2
#define PQB_TREG 24
#include "pqb_tu.inc"
