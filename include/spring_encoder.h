/*
 * include/spring_encoder.h -- C ABI of the MI355X (gfx950) encoder stage, the consumer of the
 * reorder stage's streams.
 *
 * Replaces, in memory, what spring::call_encoder -> encoder_main<N>() -> encode<N>() computes
 * (reference src/call_template_functions.cpp:65-142, src/encoder.h:124-494,:572-633,
 * src/encoder.cpp:32-109,:177-222): per contig the majority consensus (buildcontig), the
 * alignment of singleton and N reads to the consensus through two 21-base dictionaries with
 * Hamming threshold 24 (encode), the noise / position streams (writecontig), the corrected read
 * order (correct_order), the unaligned reads, and the 2-bit packing of pack_compress_seq
 * (encoder.cpp:111-156) -- everything up to, not including, the BSC calls.
 *
 * Result == the reference at `-t 1` byte for byte for the same input streams; per-tid inputs are
 * consumed tid 0, 1, ... (one legal interleaving of the reference at `-t T`).  The alignment is
 * computed by a fixed-point iteration over "first probe that takes the read" keys, which
 * reproduces the serial take order including the MAX_SEARCH_ENCODER = 1000 window of bins that
 * shrink while the scan advances.
 *
 * Return value: 0 on success, negative SPRING_REORDER_E_* on error; text in spring_reorder_last_error().
 */
#ifndef SPRING_ENCODER_H_
#define SPRING_ENCODER_H_

#include <stdint.h>

#include "spring_reorder.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spring_encoder_ctx spring_encoder_ctx;

typedef struct {
  uint64_t n_aligned;       /* stream records + aligned singletons: entries of pos / rc / noise lines */
  uint64_t n_total;         /* + unaligned reads: entries of order / readlength                       */
  uint64_t seq_len;         /* consensus bases over all contigs                                       */
  uint64_t noise_bytes;     /* read_noise.txt                                                         */
  uint64_t n_noisepos;      /* u16 entries of read_noisepos.bin                                       */
  uint64_t unaligned_bytes; /* read_unaligned.txt                                                     */
  uint64_t len_unaligned;   /* read_unaligned.txt.count                                               */
  uint64_t num_contigs;
  uint32_t matched_s, matched_N;   /* "singleton reads were aligned", "reads with N were aligned"      */
  uint32_t align_passes;    /* fixed-point passes of the alignment (1 when no bin exceeds 1000 reads) */
  uint32_t max_bin;         /* deepest singleton-dictionary bin                                       */
  double ms_device;         /* HIP-event time of the device work of the call                          */
  double ms_phase[8];       /* contigs, sort, consensus, pool+dictionaries, align, merge, noise, tail */
} spring_encoder_info;

int spring_encoder_create(int device, spring_encoder_ctx **out);
void spring_encoder_destroy(spring_encoder_ctx *ctx);
/* test hook: on = 1 makes the next encodes use one table per dictionary and the one-thread-per-window alignment kernel (what
 * runs anyway when the two dictionary windows differ in length, max_readlen <= 50) instead of the merged table.  Same output. */
int spring_encoder_set_split_tables(spring_encoder_ctx *ctx, int32_t on);

/* Encode straight from a finalized reorder context: reads and streams never leave HBM.
 * dnaN / order_N: image of input_N.dna (util.cpp:322-348 records) and read_order_N.bin
 * (preprocess.cpp:186-214), host pointers, may be NULL/0.  With NULL/0 and a reorder context that was loaded
 * through spring_reorder_load_fastq, the N reads the front end kept on the device are used (both files, merged as
 * preprocess.cpp:362-383 merges them): FASTQ text -> encoder streams without a host round trip. */
int spring_encoder_encode_reorder(spring_encoder_ctx *ctx, spring_reorder_ctx *reorder, const uint8_t *dnaN,
                                  uint64_t dnaN_bytes, const uint32_t *order_N, uint32_t numreads_N,
                                  spring_encoder_info *info);

/* The same from the in-memory images of the files encoder_main reads (encoder.h:580-593), for streams produced
 * by any reorder implementation (the reference's included):
 *   tid_count[num_thr]   records per tid; the five streams below hold the tids one after the other
 *   dna_stream           temp.dna.<tid> images concatenated (reads already reverse-complemented for 'r')
 *   order/rc/flag/pos/rlen   read_order.bin.<tid>, read_rev.txt.<tid>, tempflag.txt.<tid>, temppos.txt.<tid>,
 *                        read_lengths.bin.<tid> (decompressed)
 *   dna_single, order_s  temp.dna.singleton, read_order.bin.singleton (numreads_s records)
 *   dnaN, order_N        input_N.dna, read_order_N.bin */
int spring_encoder_encode_host(spring_encoder_ctx *ctx, uint32_t max_readlen, int32_t num_thr, const uint64_t *tid_count,
                               const uint8_t *dna_stream, uint64_t dna_bytes, const uint32_t *order, const char *rc,
                               const char *flag, const int64_t *pos, const uint16_t *rlen, const uint8_t *dna_single,
                               uint64_t single_bytes, const uint32_t *order_s, uint32_t numreads_s, const uint8_t *dnaN,
                               uint64_t dnaN_bytes, const uint32_t *order_N, uint32_t numreads_N,
                               spring_encoder_info *info);

/* Copy the streams to host buffers sized from the info struct; any pointer may be NULL:
 *   seq          seq_len chars (read_seq.bin.<tid> texts, tid-major); seq_len_tid[num_thr]
 *   pos          n_aligned u64 (read_pos.bin, absolute)
 *   noise        noise_bytes; noisepos n_noisepos u16
 *   order        n_total u32 (read_order.bin); rlen n_total u16 (read_lengths.bin); rc n_aligned
 *   unaligned    unaligned_bytes (read_unaligned.txt) */
int spring_encoder_download(spring_encoder_ctx *ctx, char *seq, uint64_t *seq_len_tid, uint64_t *pos, char *noise,
                            uint16_t *noisepos, uint32_t *order, uint16_t *rlen, char *rc, uint8_t *unaligned);

/* pack_compress_seq without BSC: per tid floor(len/4) bytes (A0 C1 G2 T3, first base in the low
 * bits) concatenated tid-major into packed, and the len%4 tail characters into tail[4*tid..]. */
int spring_encoder_download_seq_packed(spring_encoder_ctx *ctx, uint8_t *packed, char *tail);

int spring_encoder_get_info(spring_encoder_ctx *ctx, spring_encoder_info *info);

/* File contract of the two stages back to back (spring::call_reorder followed by spring::call_encoder,
 * reference src/spring.cpp:150-160): reads temp_dir/input_clean_{1,2}.dna, input_N.dna and
 * read_order_N.bin, runs the reorder chains and the encoder with everything resident in HBM (the
 * per-tid intermediate files of reorder.h:355-368 are never written), and leaves the encoder's
 * outputs in temp_dir: read_pos.bin, read_noise.txt, read_noisepos.bin, read_order.bin, read_rev.txt,
 * read_lengths.bin, read_unaligned.txt, read_unaligned.txt.count (encoder.h:365-494) and, per tid,
 * read_seq.bin.<tid>.tmp (2-bit packed) + read_seq.bin.<tid>.tail -- the state of
 * pack_compress_seq (encoder.cpp:111-156) just before its BSC_compress call, which stays with the
 * caller.  Inputs are removed like the reference does.  num_reads = clean + N reads (cp.num_reads). */
int spring_reorder_encode_run(const char *temp_dir, uint32_t max_readlen, int32_t num_thr, int32_t paired_end,
                              uint32_t num_reads_clean_1, uint32_t num_reads_clean_2, uint32_t num_reads,
                              const spring_reorder_opts *opts, spring_encoder_info *info);

/* File contract of the encoder stage alone: spring::call_encoder(temp_dir, cp) (call_template_functions.cpp:65-142)
 * up to, not including, the BSC_compress calls of pack_compress_seq (encoder.cpp:146-150): read_seq.bin.<tid> is
 * left as .tmp + .tail and the caller compresses it.  Reads the per-tid files a reorder stage left in temp_dir (the
 * reference's reorder_main or spring_reorder_run; gzip members are read through zlib), the singleton files,
 * input_N.dna and read_order_N.bin; writes the same outputs as spring_reorder_encode_run and removes its
 * inputs.  num_reads = cp.num_reads, num_reads_clean = cp.num_reads_clean[0] + cp.num_reads_clean[1]. */
int spring_encoder_run(const char *temp_dir, uint32_t max_readlen, int32_t num_thr, uint32_t num_reads,
                       uint32_t num_reads_clean, int32_t device, spring_encoder_info *info);

#ifdef __cplusplus
}
#endif
#endif
