/* oracle/orc_internal.h -- TEST INFRASTRUCTURE: types shared by reorder_oracle.c and encoder_oracle.c */
#ifndef ORC_INTERNAL_H_
#define ORC_INTERNAL_H_
#include <stdint.h>

/* bbhashdict (bitset_util.h:34-62) */
typedef struct {
  int start, end;
  uint32_t numkeys, dict_numreads;
  uint64_t *keys;     /* sorted unique (stand-in for the MPHF domain)        */
  uint32_t *startpos; /* numkeys+1 */
  uint32_t *read_id;  /* dict_numreads */
  uint8_t *empty_bin; /* numkeys */
  uint32_t *htab;     /* exact key -> bin index+1 (replaces boomphf lookup,  */
  uint64_t hmask;     /*  BooPHF.h:851; any exact map gives the same output) */
  int bpb;            /* bits per base of the bitsets: 0/2 = reorder, 3 = encoder */
} dict_t;

uint64_t orc__window64(const uint64_t *b, int W, int bitpos, int nbits);
void orc__dict_build(dict_t *d, const uint64_t *read, const uint16_t *len, uint32_t n, int W);
void orc__dict_free(dict_t *d);
int64_t orc__dict_lookup(const dict_t *d, uint64_t key);
void orc__findpos(const dict_t *d, int64_t *dictidx, uint64_t startposidx);
void orc__bin_remove(dict_t *d, int64_t *dictidx, uint64_t startposidx, int64_t current);
#endif
