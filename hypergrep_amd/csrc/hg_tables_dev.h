// Dword offsets of an automaton's tables in a staged LDS area: the confirm routines of hg_kernels.hip (one area per wave,
// CT_WORDS dwords) and the always-on tier of hg_always_on.hip (one area per workgroup, AO_TAB_WORDS dwords).
#pragma once
#include <cstdint>

#include "hg_db.h"

constexpr uint32_t CT_REACH = 0, CT_FOLLOW = 512, CT_INIT = 640, CT_AMASK = 644, CT_ACC = 676, CT_WORDS = 768;
// always-on, single-word units (<= 32 nodes): the follow step is table-driven — fu[t][b] is the union of follow[] over the set
// bits of byte t of the state word (the init nodes folded into table 0), so a step is <= 4 independent LDS reads
constexpr uint32_t CT_FU = CT_WORDS;                 // fu[4][256]
constexpr uint32_t CT_MEMBER = CT_WORDS + 1024;      // member pattern indices [8] | nodes of each [8] (a group, or one expression)
constexpr uint32_t CT_NL_ACCEPTS = CT_INIT + 3;      // != 0: a match of the staged unit can include the newline
constexpr uint32_t CT_RXA = CT_WORDS + 1024 + 2 * HG_GROUP_MAX_MEMBERS;  // lean steps: {RX, AX}[4 classes][256] (class 1 = the class of every byte), or reachL[256]
constexpr uint32_t CT_SHIFT = CT_RXA + 2048;         // shift form of the follow step: exception count (> 2: none), M1, M0, M2, src0, F0, src1, F1, M3
// (+16..: the same for a unit of two state words: flag, M0 M1 M2 M3 (lo, hi), src0, F0 (lo, hi), src1, F1 (lo, hi); its reachL2[256][2] sits at CT_RXA)
constexpr uint32_t AO_TAB_WORDS = CT_SHIFT + 32;      // the table area of hg_always_on_fast_kernel
