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
#define SYN_GENOMIC_FLAG 0x20000000u // genome with the repeat structure of a real one (syn_genomic_base)
#define SYN_FLAGS (SYN_REPEAT_FLAG | SYN_PAIRED_FLAG | SYN_GENOMIC_FLAG)
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

// ---- SYN_GENOMIC_FLAG: a genome that looks like one.  Every BASELINE config is a uniform random genome; the
// reference's slowest logged read sets are human (logs/8_29_18/NA12878-Rep-1_S1_L001.log:168: 0.14 Mreads/s at 8 threads
// against 0.83 on PhiX), where bins of thousands of reads sit beside single-read bins.  The genome is cut into
// segments of 512 bases; a segment's class comes from a hash of its index:
//   20 %  a copy of one of 64 interspersed-repeat families (512-base consensus per family).  Family sizes follow a
//         Zipf law -- on a 600 Mb genome the largest family has ~50 000 copies, the smallest ~10^3 -- and a family's copies
//         diverge from its consensus by 5, 10, 15 or 20 % substitutions (family index mod 4), drawn per copy and base;
//    5 %  a tandem repeat: a unit of 2..40 bases repeated over the segment, 2 % substitutions;
//    3 %  low complexity: one base 7 times in 8, the rest random (poly-A / AT-rich runs);
//   72 %  unique sequence, as in the uniform genome.
// A pure function of (seed, position) like everything else here: host and device emit the same bytes.
#define SYN_SEG_SHIFT 9
#define SYN_NFAM 64
SYN_HD uint32_t syn_genomic_base(uint64_t seed, uint64_t p) {
  const uint64_t sgm = p >> SYN_SEG_SHIFT;
  const uint32_t o = (uint32_t)(p & ((1u << SYN_SEG_SHIFT) - 1));
  const uint64_t hs = syn_sm64(seed ^ (0x6E0A11Cull + sgm * 0xD6E8FEB86659FD93ull));
  const uint32_t cls = (uint32_t)(hs & 0xffff);  // 16 bits: class
  // per-base randomness of this segment (substitutions, low-complexity noise)
  const uint64_t hb = syn_sm64((seed * 0x9FB21C651E98DF25ull) ^ (p >> 1) ^ 0xB45Eull);
  const uint32_t r = (uint32_t)(hb >> (32 * (p & 1)));  // 32 random bits of this base
  if (cls < 13107) {  // 20 %: interspersed repeat.  Zipf over 64 families: family f with probability ~ 1 / (f + 1)
    // f = the smallest f with cum[f] > u, cum[f] = 65536 * H_{f+1} / H_64 (harmonic numbers): an integer table, the
    // same on host and device
    const uint32_t u = (uint32_t)(hs >> 16) & 0xffff;
    static const uint16_t cum[SYN_NFAM] = {
      13815, 20722, 25327, 28781, 31544, 33846, 35820, 37547, 39082, 40463, 41719, 42870, 43933, 44920, 45841, 46704,
      47517, 48284, 49011, 49702, 50360, 50988, 51588, 52164, 52717, 53248, 53760, 54253, 54729, 55190, 55635, 56067,
      56486, 56892, 57287, 57670, 58044, 58407, 58761, 59107, 59444, 59773, 60094, 60408, 60715, 61015, 61309, 61597,
      61879, 62155, 62426, 62692, 62952, 63208, 63459, 63706, 63948, 64187, 64421, 64651, 64877, 65100, 65319, 65535};
    uint32_t f = 0;
    while (f < SYN_NFAM - 1 && cum[f] <= u) f++;
    uint32_t b = syn_genome_base(seed ^ (0xFA111ull + f * 0x9E3779B97F4A7C15ull), o);  // the family's consensus
    const uint32_t div24 = (5u + 5u * (f & 3u)) * 167772u;  // 5 / 10 / 15 / 20 % of 2^24
    if ((r & 0xFFFFFFu) < div24) b = (b + 1u + ((r >> 24) % 3u)) & 3u;
    return b;
  }
  if (cls < 16384) {  // 5 %: tandem repeat
    const uint32_t unit = 2u + (uint32_t)((hs >> 16) % 39u);
    uint32_t b = syn_genome_base(seed ^ (0x7A4Dull + sgm * 0xC2B2AE3D27D4EB4Full), o % unit);
    if ((r & 0xFFFFFFu) < 335544u) b = (b + 1u + ((r >> 24) % 3u)) & 3u;  // 2 %
    return b;
  }
  if (cls < 18350) {  // 3 %: low complexity
    const uint32_t dom = (uint32_t)(hs >> 16) & 3u;
    return (r & 7u) ? dom : ((r >> 8) & 3u);
  }
  return syn_genome_base(seed, p);
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
  const uint32_t genomic = err_thr24 & SYN_GENOMIC_FLAG;
  err_thr24 &= ~SYN_FLAGS;
  uint32_t b = genomic ? syn_genomic_base(seed, gp) : syn_genome_base(seed, gp);
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
