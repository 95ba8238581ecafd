// placeholder until the decoder lands (replaced in the next commit)
#include "kh_common.h"
using namespace kh;
#define NI() do { SetError("%s: not implemented yet", __func__); } while (0)
extern "C" {
KhFst *kh_fst_create(int32_t, int32_t, const int64_t *, const int32_t *, const int32_t *, const float *, const int32_t *, const float *) { NI(); return nullptr; }
void kh_fst_destroy(KhFst *) {}
int64_t kh_fst_num_arcs(const KhFst *) { return 0; }
void kh_decoder_config_default(KhDecoderConfig *c) { c->beam = 16.f; c->max_active = 2147483647; c->min_active = 200; c->lattice_beam = 10.f; c->prune_interval = 25; c->beam_delta = 0.5f; c->hash_ratio = 2.f; c->prune_scale = 0.1f; }
KhDecoder *kh_decoder_create(const KhFst *, const KhDecoderConfig *, int, int) { NI(); return nullptr; }
void kh_decoder_destroy(KhDecoder *) {}
int kh_decoder_decode(KhDecoder *, const float *, int, const int32_t *, int, const int32_t *) { NI(); return KH_ESTATE; }
int kh_decoder_get_stats(const KhDecoder *, int, KhDecodeStats *) { NI(); return KH_ESTATE; }
int kh_decoder_get_raw_lattice(const KhDecoder *, int, int32_t *, int32_t *, float *, int32_t *, int32_t *, int32_t *, int32_t *, float *, float *) { NI(); return KH_ESTATE; }
int kh_decoder_get_best_path(const KhDecoder *, int, int32_t *, int, int32_t *, int32_t *, int, int32_t *, float *, float *) { NI(); return KH_ESTATE; }
int kh_lattice_forward_backward(int, const int32_t *, const int64_t *, const int32_t *, const int32_t *, const float *, const float *, const float *, float *, double *, double *, int32_t *) { NI(); return KH_ESTATE; }
}
