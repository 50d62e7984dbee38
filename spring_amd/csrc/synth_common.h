// spring_amd/csrc/synth_common.h
//
// Counter-based synthetic read generator: uniform random
// genome, reads at uniform start positions, i.i.d. substitutions, 50 % reverse
// complemented, no N, no indels.  Every value is a pure function of
// (seed, index), so the host loop and the HIP kernel emit identical bytes and a
// 100 M-read set never has to exist on the host.
#ifndef SPRING_SYNTH_COMMON_H_
#define SPRING_SYNTH_COMMON_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define SYN_HD __host__ __device__ static inline
#else
#define SYN_HD static inline
#endif

#define SYN_REPEAT_FLAG 0x80000000u  // top bit of err_ppm / err_thr24: genome with 4 exact copies of one unit
#define SYN_PAIRED_FLAG 0x40000000u  // paired-end pool: read i >= n/2 is the mate of read i - n/2
#define SYN_FLAGS (SYN_REPEAT_FLAG | SYN_PAIRED_FLAG)
#define SYN_INSERT_MEAN 400          // fragment length ~ N(400, 50), clamped to [L, G]
#define SYN_INSERT_SD 50

SYN_HD uint64_t syn_sm64(uint64_t x) {  // splitmix64 finalizer
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// genome base (natural code A0 C1 G2 T3) at position p: 32 bases per hashed word
SYN_HD uint32_t syn_genome_base(uint64_t seed, uint64_t p) {
  uint64_t w = syn_sm64(seed ^ (0xA5A5A5A5ull + (p >> 5) * 0x9E3779B97F4A7C15ull));
  return (uint32_t)(w >> (2 * (p & 31))) & 3u;
}

SYN_HD uint64_t syn_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// base j (natural code) of read i, after substitution and strand flip
SYN_HD uint32_t syn_read_base(uint64_t seed, uint64_t G, uint32_t L, uint32_t err_thr24, uint64_t i,
                              uint32_t j, uint64_t pos, uint32_t rc) {
  uint32_t jj = rc ? (L - 1 - j) : j;  // position on the forward strand
  uint64_t gp = pos + jj;
  if (err_thr24 & SYN_REPEAT_FLAG) {  // "hard" genome: eighths 0,2,4,6 are exact copies of eighth 0
    const uint64_t seg = G / 8, k = seg ? gp / seg : 0;
    if (seg && k < 8 && (k & 1) == 0) gp %= seg;
  }
  err_thr24 &= ~SYN_FLAGS;
  uint32_t b = syn_genome_base(seed, gp);
  uint64_t e = syn_sm64((seed * 0x2545F4914F6CDD1Dull) ^ (i * 512ull + (jj >> 1)) ^ 0x5EEDull);
  uint32_t h = (uint32_t)(e >> (32 * (jj & 1)));
  if ((h & 0xFFFFFFu) < err_thr24) b = (b + 1u + ((h >> 24) % 3u)) & 3u;
  return rc ? 3u - b : b;
}

// start position on the forward strand and strand of read i.  npairs = 0: independent reads.  npairs > 0 (the
// pool holds 2 * npairs reads, file-1 reads then file-2 reads as reorder.h:233-242 lays them out): pair j is a
// fragment of F ~ N(400, 50) bases (sum of twelve 16-bit uniforms: integer arithmetic only, identical on host and
// device) at a uniform position on a random strand; read j is its first L bases, read npairs + j the reverse
// complement of its last L bases.
SYN_HD void syn_read_params(uint64_t seed, uint64_t G, uint32_t L, uint64_t i, uint64_t npairs, uint64_t *pos,
                            uint32_t *rc) {
  if (npairs == 0) {
    uint64_t h = syn_sm64(seed + 0x1234567ull + i * 0xD1342543DE82EF95ull);
    *pos = syn_mulhi64(h, G - L + 1);
    *rc = (uint32_t)(syn_sm64(h ^ 0xC0FFEEull) & 1u);
    return;
  }
  const uint64_t j = i >= npairs ? i - npairs : i;
  const uint32_t mate = i >= npairs ? 1u : 0u;
  uint64_t h = syn_sm64(seed + 0x1234567ull + j * 0xD1342543DE82EF95ull);
  uint64_t u0 = syn_sm64(h ^ 0xF00Dull), u1 = syn_sm64(h ^ 0xBEEFull), u2 = syn_sm64(h ^ 0xFACEull);
  long long S = 0;
  for (int k = 0; k < 4; k++) S += (long long)((u0 >> (16 * k)) & 0xffff) + (long long)((u1 >> (16 * k)) & 0xffff) + (long long)((u2 >> (16 * k)) & 0xffff);
  long long F = SYN_INSERT_MEAN + (SYN_INSERT_SD * (S - 393216)) / 65536;  // 12 uniforms: mean 6 * 65536, sd 65536
  if (F < (long long)L) F = L;
  if (F > (long long)G) F = (long long)G;
  const uint64_t p = syn_mulhi64(h, G - (uint64_t)F + 1);
  const uint32_t strand = (uint32_t)(syn_sm64(h ^ 0xC0FFEEull) & 1u);  // 1: the fragment is on the reverse strand
  // first read = fragment start on the fragment's strand; mate = other end, other strand
  const uint32_t at_end = strand ^ mate;  // 1: this read covers the last L bases of the fragment in genome coordinates
  *pos = at_end ? p + (uint64_t)F - L : p;
  *rc = at_end;
}

// natural code (A0 C1 G2 T3) -> SPRING 2-bit code (A0 G1 C2 T3, util.cpp:270-274)
SYN_HD uint32_t syn_nat_to_spring(uint32_t b) { return (b == 1u) ? 2u : (b == 2u) ? 1u : b; }

SYN_HD uint32_t syn_err_thr24(uint32_t err_ppm) {
  return (uint32_t)(((uint64_t)(err_ppm & ~SYN_FLAGS) << 24) / 1000000ull) | (err_ppm & SYN_FLAGS);
}

#endif
