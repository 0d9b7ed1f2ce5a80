// Work partition of the persistent 256-row projection GEMM (gemm_sk.hip): which (output tile, K range) segments
// each workgroup of the grid computes, and who reduces a tile that several workgroups share.
// One definition for the device code, the host launcher and the CPU tests (tests/test_host_cpu.py walks the plans
// of the step through the host-only entry fvqa_gemm_sk_describe).
//
// Vocabulary
//   wide stage  128 bytes of K per row (64 bf16 / 32 fp32): one turn of the LDS-DMA ring
//   granule     `gran` wide stages: the unit K is cut in
//   tile        256 rows x 256 columns of the output; tm x tn tiles
//   team        `ts` workgroups that walk the SAME (n tile, K range) on `ts` different m tiles at the same time:
//               they stream the same weight panel, which therefore leaves HBM once (shared through the L2 of
//               their XCD); m tiles beyond one team form further m groups
//   segment     (tile, [k0, k1) in wide stages) computed by one workgroup in one run of the ring loop
//   piece c / n the c-th of n segments a tile's K range is cut into (n == 1: the tile is computed whole)
//
// Partition (measured on MI355X, profiles/r02_gemm_partition_probe.log: an evenly dealt "stream-K" partition of the
// 172-344-tile outputs was 4-9 % SLOWER than whole tiles — under dense bf16 MFMA the chip is power-limited, CUs left idle
// by a partly filled round let the busy ones clock higher, while every extra prologue / partial-tile exchange is lost
// time — so K is split only where a round would otherwise be less than half full):
//   * `full` rounds of WHOLE tiles: team g computes tiles g, g + teams, ... (n == 1, no exchange);
//   * the `rem` tiles left over (fewer than teams) form the LAST segment of every team that gets one: each is cut into
//     s pieces (s = 1, 2, 4 or 8, the largest with s * rem <= teams and >= 2 granules per piece), one per team.
//     The s workgroups of a tile have done the same work before, finish together and share the reduction: each adds
//     128/s rows of every wave's 128 x 64 sub-tile from all s partials (fixed order 0..s-1: bitwise repeatable) and
//     stores them with the epilogue. N = 4096 outputs (64 tiles for 256 CUs) are the case full = 0, s = 4; there the
//     piece index is constant per XCD (pstride), which keeps each XCD's L2 on one K range of the activations.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define FVQA_HD __host__ __device__ __forceinline__
#else
#define FVQA_HD inline
#endif

struct fvqa_sk_plan {
  int32_t tm, tn;        // output tiles along M and N
  int32_t nw_tile;       // wide stages per tile (K / elements per wide stage)
  int32_t gran;          // wide stages per granule
  int32_t gpt;           // granules per tile = ceil(nw_tile / gran)
  int32_t ts;            // team size (m tiles walked together)
  int32_t mgroups;       // ceil(tm / ts)
  int32_t n_teams;       // teams in the grid (grid = n_teams * ts workgroups)
  int32_t full;          // rounds of whole tiles
  int32_t rem;           // tiles of the last, split round
  int32_t s;             // pieces per tile of the last round
  int32_t pstride;       // 1: the pieces of a tile on consecutive teams; > 1: teams per XCD chunk, piece index constant per chunk (fvqa_sk_piece_team)
};

struct fvqa_sk_seg {
  int32_t tile;          // tile index in walk order: n tile = tile / mgroups, m group = tile % mgroups
  int32_t k0, k1;        // wide stages [k0, k1) of the tile's K range
  int32_t n, c;          // piece c of n; piece p of the tile is held by team fvqa_sk_piece_team(plan, g, c, p)
};

FVQA_HD int fvqa_sk_tiles(const fvqa_sk_plan& p) { return p.mgroups * p.tn; }

// pstride > 1 (piece index constant per XCD chunk): which piece the chunks x = 0 .. s-1 of a group of s chunks hold. A 4-way
// split puts pieces 0, 2, 1, 3 on chunks 0, 1, 2, 3 (a self-inverse swap of the middle two): measured on the MI355X
// (profiles/r04_clock.log, per-XCD loop times on four boxes) the odd XCDs run ~4 % slower (lower clock under this load) and
// the pieces over the FIRST half of K stream slower on the long-K shapes (W1|W3^T, W2: +3..7 %); with the identity map the
// slow piece 1 sat on the slow XCDs 1 and 5 and every tile waited 6-9 us for it; this way the slow pieces run on the fast XCDs.
FVQA_HD int fvqa_sk_chunk_piece(const fvqa_sk_plan& p, int x) { return p.s == 4 ? ((x == 1 || x == 2) ? 3 - x : x) : x; }
// team that holds piece `piece` of the tile whose piece c is held by team g
FVQA_HD int fvqa_sk_piece_team(const fvqa_sk_plan& p, int g, int c, int piece) {
  if (p.pstride == 1) return g + (piece - c);
  const int x = g / p.pstride, y = g - x * p.pstride;
  return ((x / p.s) * p.s + fvqa_sk_chunk_piece(p, piece)) * p.pstride + y;        // (the chunk map is its own inverse)
}

// Segment number idx (0, 1, ...) of team g; false when the team has no such segment.
FVQA_HD bool fvqa_sk_segment(const fvqa_sk_plan& p, int g, int idx, fvqa_sk_seg* s) {
  if (idx < p.full) {
    s->tile = g + idx * p.n_teams; s->k0 = 0; s->k1 = p.nw_tile; s->n = 1; s->c = 0;
    return true;
  }
  if (idx > p.full || g >= p.rem * p.s) return false;
  int r, c;
  if (p.pstride == 1) {                                           // the pieces of a tile on consecutive teams
    r = g / p.s; c = g - r * p.s;
  } else {                                                          // piece index constant per XCD (see make_plan)
    const int x = g / p.pstride, y = g - x * p.pstride;
    c = fvqa_sk_chunk_piece(p, x % p.s); r = (x / p.s) * p.pstride + y;
  }
  const int qq = p.gpt / p.s, rr = p.gpt - qq * p.s;               // granules per piece, first rr pieces one more
  const int g0 = c * qq + (c < rr ? c : rr), g1 = g0 + qq + (c < rr ? 1 : 0);
  s->tile = p.full * p.n_teams + r;
  s->k0 = g0 * p.gran;
  s->k1 = g1 * p.gran < p.nw_tile ? g1 * p.gran : p.nw_tile;      // never empty: >= 2 granules per piece
  s->n = p.s; s->c = c;
  return true;
}

// ---- host side: choose the partition for an (M, N, K) problem on n_cu compute units
static inline fvqa_sk_plan fvqa_sk_make_plan(int M, int N, int K, int wide_elems, int n_cu) {
  fvqa_sk_plan p;
  p.tm = (M + 255) / 256;
  p.tn = (N + 255) / 256;
  p.nw_tile = K / wide_elems;
  p.gran = 4;
  p.gpt = (p.nw_tile + p.gran - 1) / p.gran;
  // teams: all m tiles of a weight panel together when they fit (<= 8), else the fewest equal m groups
  const int groups = (p.tm + 7) / 8;
  p.ts = (p.tm + groups - 1) / groups;
  p.mgroups = (p.tm + p.ts - 1) / p.ts;
  int teams = n_cu / p.ts;
  if (teams < 1) teams = 1;
  const int tiles = p.mgroups * p.tn;
  p.full = tiles / teams;
  p.rem = tiles - p.full * teams;
  p.s = 1;
  while (p.rem > 0 && p.s < 8 && p.rem * p.s * 2 <= teams && p.gpt / (p.s * 2) >= 2) p.s *= 2;
  p.n_teams = p.full > 0 ? teams : p.rem * p.s;
  // A pure split launch that fills 8 equal XCD chunks of teams (the kernel gives consecutive work ids to one XCD): every
  // team of an XCD takes the SAME piece index, so an XCD's L2 sees one K range of the activations instead of all of K
  // (measured on W1|W3^T: 4x fewer activation bytes leave L2). The partners of a tile then sit pstride teams apart, on
  // different XCDs — their exchange goes through write-through slabs wherever they sit.
  p.pstride = 1;
  if (p.full == 0 && p.s > 1 && p.n_teams % 8 == 0 && (p.n_teams * p.ts) % 8 == 0 && 8 % p.s == 0 &&
      p.rem * p.s == p.n_teams)
    p.pstride = p.n_teams / 8;
  return p;
}
