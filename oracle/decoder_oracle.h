/* decoder_oracle.h — TEST INFRASTRUCTURE. C interface of the decoder / lattice
 * restatement (decoder_oracle.cc, lattice_oracle.cc). See those files. */
#ifndef DECODER_ORACLE_H_
#define DECODER_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* fst::Fst<StdArc> as host CSR (what ReadFstKaldi yields, nnet-latgen-faster.cc:108). */
typedef struct KoFst {
  int32_t num_states;
  int32_t start;
  const int64_t *arc_offsets; /* num_states + 1 */
  const int32_t *ilabel;
  const int32_t *olabel;
  const float *weight;
  const int32_t *nextstate;
  const float *final_cost; /* +inf = not final */
} KoFst;

/* LatticeFasterDecoderConfig decoder/lattice-faster-decoder.h:40-95 */
typedef struct KoDecoderConfig {
  float beam;
  int32_t max_active;
  int32_t min_active;
  float lattice_beam;
  int32_t prune_interval;
  float beam_delta;
  float hash_ratio;
  float prune_scale;
} KoDecoderConfig;

typedef struct KoDecodeStats {
  int32_t num_frames;
  int32_t reached_final;
  float final_relative_cost;
  float final_best_cost;
  int32_t num_tokens;
  int32_t num_links;
  int64_t arcs_expanded;
  int64_t tokens_created;
  int32_t status;
  int32_t max_tokens_frame;
} KoDecodeStats;

/* mode 0: reference iteration order; mode 3: canonical (order-independent);
 * bit 1 = canonical emitting cutoff + best-token tie rule, bit 2 = canonical pruning. */
void *ko_decoder_create(const KoFst *fst, const KoDecoderConfig *cfg, int mode);
void ko_decoder_destroy(void *dec);
int ko_decoder_decode(void *dec, const float *loglikes, int T, int ll_stride, const int32_t *tid2pdf);
/* LatticeFasterOnlineDecoder call sequence (lattice-faster-online-decoder.cc:55-72,
 * 747-769,775-790): begin = InitDecoding over a decodable with num_frames_ready rows,
 * advance(max_num_frames) returns NumFramesDecoded(), finalize = FinalizeDecoding,
 * snapshot = GetRawLattice(use_final_probs) at the current point for the getters. */
int ko_decoder_begin(void *dec, const float *loglikes, int num_frames_ready, int ll_stride, const int32_t *tid2pdf);
int ko_decoder_advance(void *dec, int max_num_frames);
int ko_decoder_finalize(void *dec);
int ko_decoder_snapshot(void *dec, int use_final_probs);
int ko_decoder_get_stats(void *dec, KoDecodeStats *st);
int ko_decoder_get_raw_lattice(void *dec, int32_t *state_frame, int32_t *state_hclg, float *state_final,
                               int32_t *arc_src, int32_t *arc_dst, int32_t *arc_il, int32_t *arc_ol,
                               float *arc_g, float *arc_a);
int ko_decoder_get_best_path(void *dec, int32_t *ali, int cap_ali, int32_t *n_ali, int32_t *words, int cap_words,
                             int32_t *n_words, float *graph_cost, float *acoustic_cost);
int ko_lattice_best_path(int n_states, int n_arcs, const int32_t *arc_src, const int32_t *arc_dst,
                         const int32_t *arc_il, const int32_t *arc_ol, const float *arc_g, const float *arc_a,
                         const float *state_final, int32_t *ali, int cap_ali, int32_t *n_ali, int32_t *words,
                         int cap_words, int32_t *n_words, float *graph_cost, float *acoustic_cost);

/* lattice_oracle.cc: LatticeForwardBackward lat/lattice-functions.cc:272-354 on one
 * top-sorted lattice in CSR form; returns tot_backward_prob. */
double ko_lattice_forward_backward(int n_states, const int64_t *arc_offsets, const int32_t *arc_ilabel,
                                   const int32_t *arc_nextstate, const float *arc_graph,
                                   const float *arc_acoustic, const float *state_final, float *arc_post,
                                   double *acoustic_like_sum, int32_t *state_times, double *tot_forward);
#ifdef __cplusplus
}
#endif
#endif
