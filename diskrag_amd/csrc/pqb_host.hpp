// pqb_host.hpp -- the kernel table of DR_MODE_PQB (pqb_kernel.hpp). One translation unit per (m / 16, table rows in registers)
// pair (pqb_m<M16>_t<TREG>.hip, built in parallel by the Makefile), each instantiating list capacities 64 ... 1024 x 1 / 2 / 4
// passes per step.
#pragma once
#include "pqb_kernel.hpp"

struct PqbChoice { const void *fn; int m16, treg, nc; };
// fn[sizeclass][passes class: 1, 2, 4]
struct PqbTable { int m16, treg; const void *fn[5][3]; };
const PqbTable *dr_pqb_table_m0_t0();
const PqbTable *dr_pqb_table_m1_t0();
const PqbTable *dr_pqb_table_m1_t8();
const PqbTable *dr_pqb_table_m2_t16();
const PqbTable *dr_pqb_table_m2_t24();
const PqbTable *dr_pqb_table_m3_t16();
const PqbTable *dr_pqb_table_m3_t32();
const PqbTable *dr_pqb_table_m4_t32();
const PqbTable *dr_pqb_table_m4_t48();

// sc: size class of the list (0..4); nc: passes per step (1, 2, 4); m: sub-quantisers; treg_pref: -1 = the engine's choice
static inline PqbChoice dr_pqb_choose(int sc, int nc, uint32_t m, int treg_pref)
{
    const PqbTable *t = nullptr;
    if (m == 16) t = (treg_pref == 0 || nc > 2 || sc > 3) ? dr_pqb_table_m1_t0() : dr_pqb_table_m1_t8();     // (8 of 16 rows in registers: 8 KiB of LDS per wavefront)
    // m = 32: 24 rows in registers = 12 wavefronts per CU at 168 registers each (c5s 4M, L = 100, beam_width 8, two pops: 1.42 -> 1.27 ms
    // against 16 rows / 9 wavefronts, profiles/r05/ab/); four passes per step or 1024-entry lists do not fit 168 registers: 16 rows
    else if (m == 32) {
        // (round 6: the all-LDS form of m = 32 -- 4 wavefronts per CU, 1.6x slower -- and the 16-register-row form of m = 64 were reachable through
        //  DR_PQB_TREG only and are no longer built; m not a multiple of 16 still runs with its whole table in LDS: m0_t0)
        const int tr = treg_pref == 16 ? 16 : treg_pref == 24 ? 24 : (nc <= 2 && sc <= 3) ? 24 : 16;
        t = tr == 24 ? dr_pqb_table_m2_t24() : dr_pqb_table_m2_t16();
    }
    // m = 48 / 64 (round 6): the LDS rows set the wavefronts per CU (32 rows: 4, 16 rows: 8), so as many rows as the 256 registers of two wavefronts per
    // SIMD hold go to registers -- m = 64, L = 100, beam_width 32 on 1M x 1536: 32 / 40 / 48 rows 4.31 / 4.15 / 3.67 ms per 10 000 queries, m = 48: 16 / 32 rows 3.23 / 2.70 (and at m = 64 8.5 ms
    // before, when the 32-row form was compiled for three wavefronts per SIMD and spilled); four passes per step keep the 32-row form (no scratch)
    else if (m == 48) t = (treg_pref == 16 || (treg_pref != 32 && nc > 2)) ? dr_pqb_table_m3_t16() : dr_pqb_table_m3_t32();
    else if (m == 64) t = (treg_pref == 32 || (treg_pref != 48 && (nc > 2 || (nc == 2 && sc >= 3)))) ? dr_pqb_table_m4_t32() : dr_pqb_table_m4_t48();      // (two passes per step with lists beyond 256 entries: the 48-row form spills 900 bytes)
    else if (m <= 128) t = dr_pqb_table_m0_t0();
    PqbChoice c = { nullptr, 0, 0, 1 };
    if (!t) return c;
    const int ci = nc <= 1 ? 0 : nc <= 2 ? 1 : 2;
    c.fn = t->fn[sc][ci]; c.m16 = t->m16; c.treg = t->treg; c.nc = ci == 0 ? 1 : ci == 1 ? 2 : 4;
    return c;
}
