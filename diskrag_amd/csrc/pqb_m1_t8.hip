#define PQB_M16 1
#define PQB_TREG 8
#include "pqb_tu.inc"
