#define PQB_M16 0
#define PQB_TREG 0
#include "pqb_tu.inc"
