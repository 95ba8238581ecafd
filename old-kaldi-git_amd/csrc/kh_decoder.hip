// kh_decoder.hip — LatticeFasterDecoder on gfx950.
//
// Replaces decoder/lattice-faster-decoder.{h,cc} (Token / ForwardLink /
// HashList<StateId,Token*>, ProcessEmitting :660-750, ProcessNonemitting
// :752-812, GetCutoff :591-658, PruneActiveTokens :476-503, PruneForwardLinks
// :273-344, PruneForwardLinksFinal :349-431, PruneTokensForFrame :450-469,
// ComputeFinalCosts :505-545, FinalizeDecoding :573-588, GetRawLattice
// :109-191, GetBestPath :99-105) for a BATCH of utterances.
//
// MI355X design (see DESIGN.md "Decoder"):
//  * one 1024-thread workgroup (16 wave64) per utterance, persistent over the
//    whole utterance: every per-frame reduction / scan / select is
//    workgroup-local (LDS + barriers), no inter-workgroup communication; the
//    batch (>= 256 utterances) fills the 256 CUs;
//  * HCLG is a device CSR split into an emitting and an epsilon arc table
//    (16-byte arc records, one coalesced 16-B load per arc);
//  * tokens live in SoA arenas; the per-frame token table is an open-addressing
//    hash keyed by HCLG state with 64-bit CAS insertion and atomic-min on an
//    order-preserving integer image of the float cost; new tokens are appended
//    by wave-aggregated atomics; forward links are appended through
//    workgroup prefix sums (ballot/shuffle scans);
//  * the max-active / min-active cutoffs (std::nth_element in the reference) are
//    an exact LDS radix select over the cost images, restricted to the bits in which the
//    frame's smallest and largest key differ (2 passes of 11 bits);
//  * backward pruning iterates extra_costs to the exact fixed point per frame;
//    tokens and links live in ONE append-only arena each, in frame order; every
//    prune_interval frames the tail of the arenas (the "window") is compacted in place so
//    that the working set stays small and the slot's memory does not depend on T.
//
// Semantics: the order-independent ("canonical") resolution of the reference's
// iteration-order artefacts, defined in oracle/decoder_oracle.cc (mode 3) and
// DESIGN.md "Decoder parity".  All cost arithmetic keeps the reference's float
// association order ((cur + ac) + graph etc.); the file is compiled with
// -ffp-contract=off.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kh_common.h"

using namespace kh;

// ================================================================ device FST
// The graph as the kernel reads it: ONE table of 16-byte units in which a state is a RECORD
//   [header {#emitting arcs, first epsilon arc, #epsilon arcs, final cost}] [emitting arc] ... [emitting arc]
// and a state's id IS the unit index of its header (monotone in the caller's ids, so every smallest-
// state-id tie-break is unchanged).  Every read of this part moves a whole 128-byte line
// (profiles/r02_pmc_calibration.txt): with an offsets array + an arc table + a pdf table a token of a
// 2-arc HMM state cost three lines; its record is 48 bytes of one, next to the records of the states
// of the same HMM chain.  The epsilon arcs (3 % of the traffic) stay in their own table.
struct KhFst {
  int32_t num_states = 0, start = 0, start_state = 0;  // start: unit id; start_state: the caller's id
  int64_t num_arcs = 0, num_emit = 0, num_eps = 0, num_units = 0;
  int32_t max_emit = 0;            // the largest number of emitting arcs of one state
  int4 *rec = nullptr;             // [num_units] header {n_emit, eps_base, n_eps, final bits} | arc {ilabel, olabel, weight bits, nextstate unit | flags}
  int32_t *unit_ilabel = nullptr;  // [num_units] ilabel (> 0) of an arc unit (the decoder's copy of rec holds the pdf there); -1 - the caller's state id for a header
  int4 *n_arcs = nullptr;          // {0, olabel, weight bits, nextstate unit | flags}
  std::vector<float> final_host;       // host copy for lattice export, by the caller's state
  int start_has_eps = 0;
  int32_t max_ilabel = 0;
};

namespace {

// Every use of the lane's index in this file goes through an OPAQUE copy (round 6).  The decode kernels are one loop over
// frames around everything; what the optimizer derives from a plain threadIdx.x - `tid * 8 + j`, `tid < 16`, `~tid`, LDS
// addresses - is loop-invariant, gets computed once at the kernel's entry and stays live for the whole launch; the 64
// registers of a 1024-thread workgroup at two per CU cannot hold these values, they are spilled, and every USE becomes a
// scratch_load (a trip to memory: the scratch of 512 workgroups is 160 MB, no cache holds it) - ten such values accounted
// for 230 of the reference-order kernel's 329 static scratch loads, about 1 MB of reads per frame.  Behind an empty
// `asm volatile` the derived values are recomputed where they are used (a VALU instruction each).  Same-box A/B:
// scratch 308 -> 128 B per lane and 981 -> 903 ms (reference order), 84 -> 0 B and 526 -> 507 ms (canonical), serving
// kernel 416 -> 172 B and chunk latency p50 3.9 -> 2.8 ms.  (Round 5 had done this for the list-order routines only: OpaqueTid.)
__device__ __forceinline__ unsigned KhOpaqueTidX() {
  unsigned t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
#ifndef KH_PLAIN_TID
#define KH_TIDX (KhOpaqueTidX())
#else
#define KH_TIDX (threadIdx.x)
#endif

// -DKH_SERVE_MARKERS (debug builds): thread 0 of a serving workgroup leaves (place << 24 | detail) in its stream's control
// block in host memory as it goes; a KH_ETIMEOUT dump then says WHERE a workgroup that never came back is (round 6: one
// reference-order stream in ~3 % of the stress harness's runs hangs inside AdvanceDecoding).
#ifdef KH_SERVE_MARKERS
__device__ int32_t *g_serve_mark = nullptr;   // the ServeCtl array (64-byte blocks; word 13 = pad1[0])
#define KM(k, aux) do { if (KH_TIDX == 0 && g_serve_mark != nullptr) __hip_atomic_store(g_serve_mark + blockIdx.x * 16 + 13, ((k) << 24) | (static_cast<int32_t>(aux) & 0xffffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
// WM: the same per WAVE (lane 0 of each wave, 16 words per stream in a second host buffer): whether a stuck workgroup loops
// or waits at a barrier some of its waves never reach.  WM_ANY(c): a condition met by a lane of the wave (or-ed into bit 20).
__device__ int32_t *g_wave_mark = nullptr;
#define WM(k, aux) do { if ((KH_TIDX & 63) == 0 && g_wave_mark != nullptr) __hip_atomic_store(g_wave_mark + blockIdx.x * 64 + (KH_TIDX >> 6), ((k) << 24) | (static_cast<int32_t>(aux) & 0xffffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
#define WM_ANY(c) (__any(c) ? (1 << 20) : 0)
#else
#define KM(k, aux) do {} while (0)
#define WM(k, aux) do {} while (0)
#define WM_ANY(c) ((void)(c), 0)
#endif

#ifndef KH_NT
#define KH_NT 1024
#endif
constexpr int NT = KH_NT;          // threads per workgroup (one utterance)
constexpr int NW = NT / 64;        // waves
#ifndef KH_NPH
#define KH_NPH 160
#endif
constexpr int NPH = KH_NPH;        // diagnostic counters per slot (96 on: the fine stamps of -DKH_X_STAMPS builds)
// Arc records carry, in bit 30 of the next state, whether that state has epsilon
// arcs: a token knows it at creation without touching the graph again.
constexpr int32_t kHasEps = 0x40000000, kStateMask = 0x1fffffff;
// bit 29: the state is the destination of some epsilon arc, i.e. the epsilon closure may look
// it up - only such tokens are entered in the global hash table (the emitting pass dedupes in LDS)
constexpr int32_t kEpsDst = 0x20000000;
constexpr int EU = 2;              // chunks of NT tokens an expansion group scans per barrier (2: -1 %; 4 spills)
constexpr int KC = 4;              // chunks of NT slots the compaction moves per barrier when the slide has opened a gap
constexpr int PU = 1;              // token / link slots a lane keeps in flight per round of a sweep (measured: 1 beats 2, 4, 8 - the sweeps are bound by the CU's address pipeline, not by latency, and more slots spill)
constexpr uint32_t kEncInf = 0xFF800000u;  // Enc(+inf)
constexpr unsigned long long kEmpty = 0ull;

__host__ __device__ __forceinline__ uint32_t Enc(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float Dec(uint32_t e) {
  uint32_t u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
  return __builtin_bit_cast(float, u);
}

// Device pointers kept in structs are declared in address space 1 (global): a plain
// `T *` loaded from memory is a GENERIC pointer to hipcc, which then emits
// flat_load / flat_store (address-space check per access, counted on vmcnt AND
// lgkmcnt) and — MI355X guide, Guideline 16 — an sc1 flat_ access is not a
// dependable L1 bypass.  With the address space in the type every access below is
// a global_* instruction.
#define GP(T) __attribute__((address_space(1))) T *

// A global array addressed with a 32-BIT UNSIGNED BYTE OFFSET: a[i] is
// *(base + zext(uint32(i) * sizeof(T))), exactly the "SGPR base + 32-bit VGPR offset"
// form of global_load / global_store / global_atomic.  With plain pointers and int
// indices every access sign-extends and shifts to a 64-bit address in a VGPR pair of
// its own (464 of the kernel's 579 loads did), which is what pushed the sweeps over
// 64 VGPRs; here the six SoA arrays of a link sweep share ONE offset register.
// Precondition (checked on the host): count * sizeof(T) < 4 GiB.
template <class T>
struct Arr {
  GP(T) p;
  Arr() = default;
  __host__ __device__ Arr(GP(T) q) : p(q) {}
  template <class U>
  __host__ __device__ Arr(const Arr<U> &o) : p(o.p) {}
  __device__ __forceinline__ __attribute__((address_space(1))) T &operator[](int i) const {
    return *(GP(T))((__attribute__((address_space(1))) char *)p + static_cast<uint32_t>(i) * static_cast<uint32_t>(sizeof(T)));
  }
  __device__ __forceinline__ __attribute__((address_space(1))) T &operator[](uint32_t i) const {
    return *(GP(T))((__attribute__((address_space(1))) char *)p + i * static_cast<uint32_t>(sizeof(T)));
  }
  __host__ __device__ operator GP(T)() const { return p; }
};
// 16-byte arc record as a native vector (HIP's int4 is a class; it cannot be
// loaded through an address-space-qualified pointer)
typedef int KhInt4 __attribute__((ext_vector_type(4)));
// Four consecutive 4-byte elements of an SoA array with ONE 16-byte access per lane, at any 4-byte
// alignment: a coalesced dword-per-lane sweep streams at 3.3 TB/s on this part, the same sweep with 16
// bytes per lane at 5.6 TB/s (profiles/r02_pmc_calibration.txt: CalStreamDword / CalStreamDwordx4).
typedef int KhInt4U __attribute__((ext_vector_type(4), aligned(4)));
typedef float KhFloat4 __attribute__((ext_vector_type(4)));
typedef float KhFloat4U __attribute__((ext_vector_type(4), aligned(4)));
template <class T>
__device__ __forceinline__ KhInt4 Load4I(const Arr<T> &a, int i) {
  static_assert(sizeof(T) == 4, "4-byte elements");
  return *(__attribute__((address_space(1))) const KhInt4U *)((__attribute__((address_space(1))) const char *)a.p + static_cast<uint32_t>(i) * 4u);
}
template <class T>
__device__ __forceinline__ KhFloat4 Load4F(const Arr<T> &a, int i) {
  static_assert(sizeof(T) == 4, "4-byte elements");
  return *(__attribute__((address_space(1))) const KhFloat4U *)((__attribute__((address_space(1))) const char *)a.p + static_cast<uint32_t>(i) * 4u);
}
template <class T>
__device__ __forceinline__ void Store4I(const Arr<T> &a, int i, KhInt4 v) {
  static_assert(sizeof(T) == 4, "4-byte elements");
  *(__attribute__((address_space(1))) KhInt4U *)((__attribute__((address_space(1))) char *)a.p + static_cast<uint32_t>(i) * 4u) = v;
}
// Non-temporal forms (the nt cache policy: the line is not kept in L2 for this access): for the arrays a frame writes once
// and nobody reads before the next pruning visit - far behind in the stream - and for that visit's reads.  What they no
// longer displace is the part of the arc table the next frames re-read (the active states of consecutive frames overlap).
template <class T>
__device__ __forceinline__ KhInt4 Load4I_NT(const Arr<T> &a, int i) {
  static_assert(sizeof(T) == 4, "4-byte elements");
  return __builtin_nontemporal_load((__attribute__((address_space(1))) const KhInt4U *)((__attribute__((address_space(1))) const char *)a.p + static_cast<uint32_t>(i) * 4u));
}
template <class T>
__device__ __forceinline__ KhFloat4 Load4F_NT(const Arr<T> &a, int i) {
  static_assert(sizeof(T) == 4, "4-byte elements");
  return __builtin_nontemporal_load((__attribute__((address_space(1))) const KhFloat4U *)((__attribute__((address_space(1))) const char *)a.p + static_cast<uint32_t>(i) * 4u));
}
template <class T>
__device__ __forceinline__ void Store4I_NT(const Arr<T> &a, int i, KhInt4 v) {
  static_assert(sizeof(T) == 4, "4-byte elements");
  __builtin_nontemporal_store(v, (__attribute__((address_space(1))) KhInt4U *)((__attribute__((address_space(1))) char *)a.p + static_cast<uint32_t>(i) * 4u));
}
template <class T>
__device__ __forceinline__ void Store4F(const Arr<T> &a, int i, KhFloat4 v) {
  static_assert(sizeof(T) == 4, "4-byte elements");
  *(__attribute__((address_space(1))) KhFloat4U *)((__attribute__((address_space(1))) char *)a.p + static_cast<uint32_t>(i) * 4u) = v;
}

// Token costs are updated with L2 atomics (atomicMin), which do not refresh this
// CU's vector L1: a plain load could return a stale L1 copy of the line (e.g. one
// fetched while reading the previous frame's tokens that share it).  Every read
// of tok_cost therefore goes to L2 (sc1 load).
template <class P>
__device__ __forceinline__ float LoadExtra(P p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class P>
__device__ __forceinline__ void StoreExtra(P p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class P>
__device__ __forceinline__ uint32_t LoadCostEnc(P p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Per-slot temporaries of the exact reference order (see "exact reference order" below).
struct UttX {
  // ---- exact reference order (Params::exact_order; carved only then).  [tok_frame_cap] unless noted.
  Arr<int32_t> x_pos;      // frontier token (i - frame begin) -> its position in the reference's HashList order
  Arr<uint32_t> x_m;       // by list position: Enc(min tot_cost + adaptive_beam) over the token's emitting arcs; after the scan the running next_cutoff BEFORE the token
  Arr<int32_t> x_c;        // by list position: # emitting arcs expanded; after the scan their exclusive prefix sum (ordinal of the token's first candidate)
  Arr<uint32_t> x_q;       // token of the frame under construction (i - nb) -> insertion key (order of HashList::Insert calls)
  Arr<int32_t> x_bkt;      // ... -> HashList bucket (caller's state id % hash size)
  Arr<int32_t> x_epsidx;   // ... -> index in tmp_epslist, or -1
  Arr<int32_t> x_nl0; Arr<int32_t> x_nl1;   // closure replay, by tmp_epslist index: the token's epsilon link slots [l0, l1) relative to the block
  Arr<float> x_ncost;      // closure replay: token cost as the replay proceeds
  Arr<int32_t> x_ord;      // [link_frame_cap] candidate ordinal of a materialised emitting candidate; then the closure replay's link destination codes
  Arr<float> x_lw;         // [link_frame_cap] closure replay: link weight
  Arr<int32_t> x_stack;    // [link_frame_cap] closure replay: the LIFO queue (:766-811)
  Arr<uint32_t> x_bmin;    // [x_hcap] HashList bucket -> smallest insertion key in it (all ones = empty: invariant between frames)
  Arr<unsigned long long> x_key0; Arr<unsigned long long> x_key1;   // radix sort keys, double buffered
  Arr<int32_t> x_val0; Arr<int32_t> x_val1;                         // radix sort payload
  Arr<int32_t> x_h; Arr<int32_t> x_inb;   // token of the frame under construction -> insertion rank of its bucket's first token; its rank inside the bucket
  Arr<uint32_t> x_c0e;     // entry of tmp_epslist made by the emitting pass -> the token's cost image before the epsilon closure
  Arr<int32_t> x_csid;     // [link_frame_cap] materialised candidate -> the caller's id of its destination state (what the reference hashes)
  Arr<const KhInt4> x_rec0;   // the graph's own records (KhFst::rec): in this mode the decoder's copy carries the caller's id of an
                              // arc's destination state where the output label was, and the export reads the label from here
  int32_t x_hcap;
};

// An array of UttX (constant address space: the pointer is fetched with a scalar load where it is used)
template <class T>
__device__ __forceinline__ Arr<T> XArr(const __attribute__((address_space(4))) Arr<T> &a) { return Arr<T>(a.p); }
#define UX(field) XArr(sh.x->field)

// Per-utterance arenas and parameters (device-resident array of these).
//
// Tokens and links live in one append-only arena each, in frame order:
//   tokens: frame 0, frame 1, ...          links: eps(0), emit(0), eps(1), emit(1), ...
// Every prune_interval frames the tail of both arenas — the "window" = all frames
// since the compaction before the previous one — is compacted in place (sliding
// the survivors down).  Frames older than the window have been compacted twice
// (the second time >= prune_interval frames behind the frontier, i.e. already
// thinned to lattice density) and never move again.
struct Utt {
  // inputs
  GP(const float) ll;   // first row of this utterance's log-likelihood matrix
  int32_t ll_stride, T;
  // token arena
  int32_t tok_cap;
  Arr<int32_t> tok_state;    // HCLG state, -1 = pruned token
  Arr<uint32_t> tok_cost;    // Enc(tot_cost); free slots hold Enc(+inf)
  Arr<float> tok_extra;
  // link arena
  int32_t link_cap;
  Arr<int32_t> link_dst; Arr<int32_t> link_arc;   // dst: token index, -1 = excised; arc: index of the emitting arc, or -1 - index of the epsilon arc (labels are read from the arc at export)
  Arr<int32_t> link_src;     // owning token (links are also walked link-parallel)
  // link_k: the part of link_extra_cost (:309-311) that does not depend on extra_costs: for an emitting link the
  // candidate's tot_cost = (cost[src] + acoustic) + graph until the frame's first pruning visit, which turns it
  // into tot_cost - cost[dst] (cost[dst] is final by then); for an epsilon link that difference from its creation
  // on.  link_a: acoustic cost (emitting links only).  The graph cost of a link is its arc's weight (read at export).
  Arr<float> link_k; Arr<float> link_a;
  // per-frame bookkeeping
  Arr<int32_t> frame_b; Arr<int32_t> frame_e;      // [T+2] token range of frame f
  Arr<int32_t> feps_b; Arr<int32_t> feps_e;        // [T+2] link range of eps(f)
  Arr<int32_t> femit_b; Arr<int32_t> femit_e;      // [T+2] link range of emit(f)
  Arr<float> cost_offset;    // [T+1]
  Arr<uint8_t> must_links;   // [T+2] must_prune_forward_links
  Arr<uint8_t> must_toks;    // [T+2] must_prune_tokens
  // temporaries
  Arr<int32_t> tmp_slot;     // [tok_frame_cap] hash slot of frontier token (i - frontier begin)
  Arr<int32_t> tmp_dirty;    // [tok_frame_cap] 1 = queued in a nonemitting work list; all zero outside ProcessNonemitting
  Arr<int32_t> tmp_work0; Arr<int32_t> tmp_work1;  // [tok_frame_cap] nonemitting work lists (token indices), double buffered
  Arr<int32_t> tmp_epslist;  // [tok_frame_cap] the frontier's tokens whose state has epsilon arcs (each once, in creation order)
  Arr<float> tmp_f0;         // [tok_frame_cap] prune: extra_cost on entry (i - frame begin)
  Arr<uint32_t> tmp_acc0; Arr<uint32_t> tmp_acc1;  // [tok_frame_cap] prune: Enc(min link_extra_cost) over emitting / epsilon links
  Arr<int32_t> tmp_remap;    // [window_cap] compaction remap (i - window begin)
  int32_t tok_frame_cap, link_frame_cap, window_cap;
  // survivors of FinalizeDecoding (lazy schedule): {token index, frame} / {link slot, frame} pairs, what ExportSurvivors copies
  Arr<int32_t> surv_tok; Arr<int32_t> surv_link;
  int32_t surv_tok_cap, surv_link_cap;
  // hash
  Arr<unsigned long long> hash;
  uint32_t hash_mask;
  GP(long long) phase_cycles;  // [16] diagnostic (KH_DECODER_PROFILE=1), else nullptr
};

struct Params {
  Arr<const KhInt4> rec;      // the decoder's copy of KhFst::rec with the pdf in the first word of every arc unit
  Arr<const KhInt4> n_arcs;
  Arr<const int32_t> unit_ilabel;
  int32_t start, num_units, num_eps, start_has_eps;
  int32_t max_emit;           // largest emitting fan-out of a state (reference order: 16-bit arc counts in LDS when it fits)
  int32_t ll_cols;  // > 0: columns of the log-likelihood matrix, staged per frame in LDS
  int32_t keep_ac;  // 1: links store their acoustic cost (online decoding: a chunk's scores are gone when the lattice is
                    // exported); 0: it is recomputed at export from the score matrix, cost_offset[f] - loglike(f, pdf of the
                    // arc) - the same float expression - and the expansion writes one stream less per candidate
  int32_t max_tid;
  // 1 (offline batch decoding): the backward pruning between frames runs only when a slot's arenas are about to fill up
  // (a garbage collection), and FinalizeDecoding prunes every frame - most of them for the first and only time.  Under
  // the canonical rule P (exact fixed point, then excise) the final lattice does not depend on WHEN the intermediate
  // PruneActiveTokens calls run: an extra_cost computed against a frontier at frame t is a lower bound of the one
  // FinalizeDecoding computes (the frontier's own extra_costs, 0 at the time, only grow; float addition and min are
  // monotone), so a link excised early is excised at the end as well, and the final sweep recomputes every surviving
  // token's extra_cost from scratch.  0: PruneActiveTokens every prune_interval frames as :88-89 (online decoding, whose
  // mid-utterance getters expose that state).
  int32_t lazy_prune;
  // lazy schedule, online streams (round 6): a garbage collection also once this many frames have gone unpruned, not only
  // when the arenas run low - the collection of a 2000-frame backlog in the middle of an utterance was a 100 - 500 ms
  // stall of that stream's chunk (KH_SERVE_LAZY_SPAN frames; 0 = arenas only, the offline kernel's rule and the DEFAULT:
  // see OnlineLazySpan).  The lattice does not depend on when the collections run.
  int32_t lazy_span;
  // 1: the reference's iteration order is reproduced (HashList order, running next_cutoff, first-minimum tie, the LIFO
  // order of the epsilon closure's insertions) - see "exact reference order" below; the kernels are instantiated for it
  int32_t exact_order;
  // epsilon closure in LDS (ClosureLds): entries its table may take (<= kClMaxLoad; 0 = every frame takes the general
  // routine).  KH_DECODER_CLOSURE_CAP lowers it: the tests run whole suites through the general routine (0) and through
  // tables that fill up on the way (a handful of entries).
  int32_t cl_max_load;
  float hash_ratio;
  float beam, lattice_beam, beam_delta, prune_scale;
  int32_t max_active, min_active, prune_interval;
};

// SGPR hygiene.  `Utt u = slots[...]` and the by-value `Params` arrive through s_load_dwordx8 / x16: the fields of one
// load form ONE 256- / 512-bit register tuple, and when the allocator runs out of scalar registers (these two structs
// alone hold ~60 pointers) it spills and reloads such a tuple AS A WHOLE - the round-4 listing of DecodeKernel<1,0> had
// 5 700 v_readlane_b32 among its 22 k instructions (a quarter of them, every one a VALU issue slot), e.g. sixteen of them
// in front of a single global_load inside the expansion loops to recover one pointer.  An empty asm with a "+s" operand
// per field ends the tuple's life at kernel entry: every field becomes a scalar value of its own, which is kept,
// spilled (two lanes of a VGPR) and reloaded (two v_readlane) on its own.
// (A tied "+s" operand is coalesced straight back into the tuple; a real s_mov inside the asm gives the copy a register
// of its own.  ~60 scalar moves per workgroup, once.)
template <class T>
__device__ __forceinline__ void LaunderOne(T &x) {
  T y;
  if constexpr (sizeof(T) == 8) asm volatile("s_mov_b64 %0, %1" : "=s"(y) : "s"(x));
  else asm volatile("s_mov_b32 %0, %1" : "=s"(y) : "s"(x));
  x = y;
}
#ifndef KH_NO_LAUNDER
#define KH_LAUNDER(x) LaunderOne(x)
#else
#define KH_LAUNDER(x) do {} while (0)
#endif
// per-phase switches of the local copies (same-box A/B builds)
#ifdef KH_NO_P2_LOCAL
#define KH_LAUNDER_P2(x) do {} while (0)
#else
#define KH_LAUNDER_P2(x) KH_LAUNDER(x)
#endif
#ifdef KH_NO_GC_LOCAL
#define KH_LAUNDER_GC(x) do {} while (0)
#else
#define KH_LAUNDER_GC(x) KH_LAUNDER(x)
#endif
#ifdef KH_NO_NE_LOCAL
#define KH_LAUNDER_NE(x) do {} while (0)
#else
#define KH_LAUNDER_NE(x) KH_LAUNDER(x)
#endif
#ifdef KH_NO_FB_LOCAL
#define KH_LAUNDER_FB(x) do {} while (0)
#else
#define KH_LAUNDER_FB(x) KH_LAUNDER(x)
#endif
#ifdef KH_NO_X_LOCAL
#define KH_LAUNDER_X(x) do {} while (0)
#else
#define KH_LAUNDER_X(x) KH_LAUNDER(x)
#endif
__device__ __forceinline__ void Launder(Utt &u) {
  KH_LAUNDER(u.ll); KH_LAUNDER(u.ll_stride); KH_LAUNDER(u.T); KH_LAUNDER(u.tok_cap);
  KH_LAUNDER(u.tok_state.p); KH_LAUNDER(u.tok_cost.p); KH_LAUNDER(u.tok_extra.p);
  KH_LAUNDER(u.link_cap);
  KH_LAUNDER(u.link_dst.p); KH_LAUNDER(u.link_arc.p); KH_LAUNDER(u.link_src.p); KH_LAUNDER(u.link_k.p); KH_LAUNDER(u.link_a.p);
  KH_LAUNDER(u.frame_b.p); KH_LAUNDER(u.frame_e.p); KH_LAUNDER(u.feps_b.p); KH_LAUNDER(u.feps_e.p);
  KH_LAUNDER(u.femit_b.p); KH_LAUNDER(u.femit_e.p); KH_LAUNDER(u.cost_offset.p);
  KH_LAUNDER(u.must_links.p); KH_LAUNDER(u.must_toks.p);
  KH_LAUNDER(u.tmp_slot.p); KH_LAUNDER(u.tmp_dirty.p); KH_LAUNDER(u.tmp_work0.p); KH_LAUNDER(u.tmp_work1.p);
  KH_LAUNDER(u.tmp_epslist.p); KH_LAUNDER(u.tmp_f0.p); KH_LAUNDER(u.tmp_acc0.p); KH_LAUNDER(u.tmp_acc1.p); KH_LAUNDER(u.tmp_remap.p);
  KH_LAUNDER(u.tok_frame_cap); KH_LAUNDER(u.link_frame_cap); KH_LAUNDER(u.window_cap);
  KH_LAUNDER(u.surv_tok.p); KH_LAUNDER(u.surv_link.p); KH_LAUNDER(u.surv_tok_cap); KH_LAUNDER(u.surv_link_cap);
  KH_LAUNDER(u.hash.p); KH_LAUNDER(u.hash_mask);
}
__device__ __forceinline__ void Launder(Params &p) {
  KH_LAUNDER(p.rec.p); KH_LAUNDER(p.n_arcs.p); KH_LAUNDER(p.unit_ilabel.p);
  KH_LAUNDER(p.start); KH_LAUNDER(p.num_units); KH_LAUNDER(p.num_eps); KH_LAUNDER(p.start_has_eps); KH_LAUNDER(p.ll_cols);
  KH_LAUNDER(p.keep_ac); KH_LAUNDER(p.max_tid); KH_LAUNDER(p.lazy_prune); KH_LAUNDER(p.exact_order); KH_LAUNDER(p.cl_max_load);
  KH_LAUNDER(p.hash_ratio); KH_LAUNDER(p.beam); KH_LAUNDER(p.lattice_beam); KH_LAUNDER(p.beam_delta); KH_LAUNDER(p.prune_scale);
  KH_LAUNDER(p.max_active); KH_LAUNDER(p.min_active); KH_LAUNDER(p.prune_interval);
}

// Workgroup barrier that also waits for this wave's outstanding vector-memory
// operations.  hipcc's __syncthreads() is a WORKGROUP-scope fence: on gfx950 (one
// CU, shared L1) it does not wait for global stores to be performed at L2.  This
// kernel communicates between waves partly through L2 (atomics, sc1 loads/stores
// of words that atomics update), so a store issued before the barrier must have
// reached L2 before another wave's L2 read after it: s_waitcnt vmcnt(0) first.
#ifdef KH_BOUNDS_CHECK
__device__ int g_oob[8];
#define KH_BOUND(code, v, lo, hi) do { if ((v) < (lo) || (v) >= (hi)) { if (atomicAdd(&g_oob[0], 1) == 0) { g_oob[1] = (code); g_oob[2] = (int)(v); g_oob[3] = (int)(lo); g_oob[4] = (int)(hi); g_oob[5] = KH_TIDX; } (v) = (lo); } } while (0)
#else
#define KH_BOUND(code, v, lo, hi) do {} while (0)
#endif
#ifdef KH_BARRIER_CHECK
__device__ int g_bar_cnt[256 * 16];
__device__ int g_bar_misaligned[4];
#endif
__device__ __forceinline__ void KhSync() {
#ifdef KH_BARRIER_CHECK
  const int w_ = KH_TIDX >> 6;
  int *bc_ = g_bar_cnt + blockIdx.x * 16;
  if ((KH_TIDX & 63) == 0) __hip_atomic_fetch_add(&bc_[w_], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef KH_BARRIER_CHECK
  {
    const int n_ = __hip_atomic_load(&bc_[w_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), v_ = __hip_atomic_load(&bc_[(w_ + 1) & (NW - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v_ != n_ && v_ != n_ + 1) {
      if (atomicAdd(&g_bar_misaligned[0], 1) == 0) { g_bar_misaligned[1] = n_; g_bar_misaligned[2] = v_; g_bar_misaligned[3] = w_; }
    }
  }
#endif
}

// Barrier for hand-offs through LDS only: no wait for this wave's outstanding global stores.
__device__ __forceinline__ void LdsSync() { __syncthreads(); }

// Workgroup-uniform values that reach a lane through LDS or a vector load sit in a
// VGPR (the compiler cannot know they are uniform) and make every loop bound and branch
// that depends on them a vector one.  Uni() moves such a value to an SGPR
// (v_readfirstlane): only for values that are uniform BY CONSTRUCTION.
__device__ __forceinline__ int Uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t Uni(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }
__device__ __forceinline__ float Uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ unsigned long long Uni(unsigned long long v) {
  return (static_cast<unsigned long long>(Uni(static_cast<uint32_t>(v >> 32))) << 32) | Uni(static_cast<uint32_t>(v));
}
__device__ __forceinline__ long long Uni(long long v) { return static_cast<long long>(Uni(static_cast<unsigned long long>(v))); }

// ---------------------------------------------------------------- block helpers
struct Shared {
  int kcnt[3][NW];                 // (reference-order kernels) call counters of the block primitives, one per wave: Blk::lds_cnt
  int wsum[2][NW];                 // BlockExScan, double buffered
  int wsumk[2][8][NW];             // BlockExScanK (K <= 8), double buffered
  int wl_n[3];                     // nonemitting work-list lengths, rotating ([0] also: prune's epsilon-token list)
  int pr_moved;                    // PruneForwardLinks: tokens whose extra_cost moved by more than delta
  int eps_n;                       // length of tmp_epslist
  unsigned long long wred[2][NW];  // block reductions, double buffered
  int orbuf[4];                    // BlockOr / BlockAny, 4 rotating slots
  unsigned long long wmin[NW];
  int flag;
  int bcast_i[4];
  float bcast_f[8];
  unsigned int hist[1 << 11];   // RadixSelect digit histogram (kRadixBits).  (NOT alignas(16): same-box A/B in round 5, the canonical kernel 554 -> 603 ms with the
                                    // array moved to a 16-byte boundary - every LDS array behind it moves with it)
  int ex_off[EU * NT], ex_ab[EU * NT], ex_tok[EU * NT];  // ExpandSweep: first link slot, first arc, token of the group's items
  // running state (owned by thread 0, read after barriers)
  int tok_end, link_end;
  int link_cursor;  // ExpandSweepFiltered: next free link slot (LDS atomic, one add per wave)
  int work_cursor;  // ExpandWavesFiltered: next unclaimed token
  uint32_t bound_enc;  // ExpandWavesFiltered: Enc(upper bound of the final next_cutoff), atomic min
  float wbound[2][NW];  // ExpandSweepFiltered: per-wave minima of the cutoff estimate, double buffered
  int own[NW][64];      // ExpandWavesFiltered: per wave, arc slot of the current 64-arc batch -> (batch tag, lane of the token that starts there)
  int front_b;  // first token of the frame under construction (frontier)
  int conv_upto;  // frames below it have had their first pruning visit (link_k of their emitting links is converted)
  int status;
  long long arcs_expanded, tokens_created;
  int max_tokens_frame;
  long long t_last, t_sub;
  long long phase[NPH];
  int tok_hw;  // highest token slot dirtied by this slot's utterances so far
  int gc_tok, gc_link;  // arena ends right after the last full compaction (garbage collection)
  int surv_nt, surv_nl; // FinalBackward: survivors listed so far (tokens, links)
  int map_n;            // FinalBackward: entries in the hand-off map (survivors of the frame just visited)
  int sched[4];         // lazy schedule, per utterance: garbage collections, dense / general final visits, hand-offs through memory
  long long cand_mat;   // emitting candidates that got a link slot (materialised), per utterance
  // exact reference order
  uint32_t x_hsize;     // HashList::hash_size_ (:219-225: grows to hash_ratio x the frame's token count, never shrinks)
  uint32_t x_qbase;     // first insertion key of the epsilon closure's insertions (> every emitting candidate's ordinal)
  int x_nbp;            // reference order, frames of 8 k - 16 k tokens: steps of the running cutoff found by the scan
  int x_ne_emit;        // end of the tokens the emitting pass created (the closure's follow)
  int x_eps_emit;       // entries of tmp_epslist the emitting pass made
  int x_n_new;          // closure replay: insertions counted
  uint32_t x_cb[16];    // sweep of the emitting arcs: Enc(upper bound of the running next_cutoff) in front of the k-th sixteenth of the list
  // epsilon closure in LDS (ClosureLds)
  int erel_n;           // entries of the frame's list of epsilon-relevant tokens (pass 2 appends; tmp_work0 / tmp_work1)
  int cl_n;             // entries of the LDS closure table
  int hash_dirty;       // 1: the global hash holds entries of the frontier (frames that took the general closure path)
};

// Per-thread view of the workgroup state: the LDS block plus the (uniform)
// rotation counters of the barrier-light block primitives below.
typedef __attribute__((address_space(3))) Shared LdsShared;
struct Blk {
  LdsShared *p;
  __attribute__((address_space(3))) float *ll_row;  // the frame's log-likelihood row staged in LDS (ll_cols > 0)
  // exact reference order: the slot's temporaries (UttX), in constant memory, fetched where they are used.  NOT a member
  // of Utt: the kernels hold that struct in registers for the whole launch, and one more pointer in it cost the canonical
  // kernel 5 % (scratch 168 -> 184 bytes per lane; same-box A/B against the round-3 build)
  __attribute__((address_space(4))) const struct UttX *x;
  int k_or, k_red, k_scan;
  // The call counters that pick a primitive's buffer.  In registers they are three values that live for the whole launch;
  // the reference-order kernels, already short of registers, kept them in scratch and paid a load + store (two trips to
  // memory) per block primitive - half of the kernel's remaining scratch traffic in round 6 (67 of 138 static scratch
  // instructions).  lds_cnt (a compile-time constant per kernel): one counter per WAVE in LDS instead, read by the wave and
  // bumped by its active lanes (all write the same value) - the LDS unit executes a wave's operations in order, and every wave makes the same calls.
  bool lds_cnt;
  __device__ __forceinline__ LdsShared *operator->() const { return p; }
};

// which: 0 = BlockOr's slot counter, 1 = the reductions', 2 = the scans'.  Returns the counter's value before this call.
template <int kWhich>
__device__ __forceinline__ int NextCall(Blk &sh) {
  if (sh.lds_cnt) {
    const int w = static_cast<int>(KH_TIDX >> 6);
    const int v = sh->kcnt[kWhich][w];
    sh->kcnt[kWhich][w] = v + 1;   // (every ACTIVE lane, the same value: a call with lane 0 masked off still counts)
    return v;
  }
  if (kWhich == 0) return sh.k_or++;
  if (kWhich == 1) return sh.k_red++;
  return sh.k_scan++;
}
// (at the kernel's entry, in front of a barrier)
__device__ __forceinline__ void InitCalls(Blk &sh, bool lds_cnt) {
  sh.k_or = 0;
  sh.k_red = 0;
  sh.k_scan = 0;
  sh.lds_cnt = lds_cnt;
  if (lds_cnt && KH_TIDX < 3 * NW) (&sh->kcnt[0][0])[KH_TIDX] = 0;
}

// The emitting pass dedupes the frame's new tokens in an LDS table (state -> min cost ->
// token index) of 8192 slots laid over LDS that is idle at that point of the frame: the
// keys over the radix-select histogram + the three expansion arrays (32 KB of the static
// block), the values over the frame's score row (dynamic LDS, at least 32 KB; the row is
// staged again at the start of the next frame).
constexpr int kLdsSlots = 8192;
static_assert(sizeof(unsigned int) * (1 << 11) + 3 * sizeof(int) * EU * NT >= sizeof(uint32_t) * kLdsSlots,
              "the LDS token table's keys are laid over hist + ex_off + ex_ab + ex_tok");
static_assert(offsetof(Shared, ex_off) == offsetof(Shared, hist) + sizeof(unsigned int) * (1 << 11) &&
              offsetof(Shared, ex_ab) == offsetof(Shared, ex_off) + sizeof(int) * EU * NT &&
              offsetof(Shared, ex_tok) == offsetof(Shared, ex_ab) + sizeof(int) * EU * NT, "contiguous LDS arrays");
__device__ __forceinline__ __attribute__((address_space(3))) uint32_t *LdsKeys(const Blk &sh) {
  return (__attribute__((address_space(3))) uint32_t *)&sh.p->hist[0];
}
__device__ __forceinline__ __attribute__((address_space(3))) uint32_t *LdsVals(const Blk &sh) {
  return (__attribute__((address_space(3))) uint32_t *)sh.ll_row;
}
// dynamic LDS of the decode kernels: the score row or the table's values, whichever is larger
inline size_t DynLdsBytes(int ll_cols) { return std::max(sizeof(float) * static_cast<size_t>(ll_cols), sizeof(uint32_t) * static_cast<size_t>(kLdsSlots)); }

// Diagnostic phase timer: thread 0 charges the shader cycles since the previous
// stamp to `ph`.  Only active when the host passed a phase_cycles buffer.
__device__ __forceinline__ void Stamp(const Utt &u, Blk &sh, int ph) {
  if (u.phase_cycles != nullptr && KH_TIDX == 0) {
    const long long now = static_cast<long long>(__builtin_amdgcn_s_memtime());
    sh->phase[ph] += now - sh->t_last;
    sh->t_last = now;
  }
}

// ---------------------------------------------------------------- wave primitives
// Cross-lane scans as DPP modifiers of VALU instructions (row_shr:1/2/4/8 inside the 16-lane rows, then row_bcast:15 and
// row_bcast:31 across them - the gfx9 wave scan): six VALU instructions and nothing on the LDS pipe.  A __shfl_up step is
// a ds_bpermute_b32 (an LDS-pipe instruction with its round trip) + the lane address + a select under a lane mask that
// the compiler keeps in a scalar register pair - and, this kernel being short of scalar registers, reloads from a spill
// VGPR with two v_readlane in front of every step.  KH_NO_DPP keeps the shuffles (same-box A/B).
#ifndef KH_NO_DPP
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int DppMov(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, kCtrl, kRowMask, 0xf, false);
}
#define KH_DPP_SCAN(v, id, OP)                  \
  v = OP(v, DppMov<0x111, 0xf>(id, v));         \
  v = OP(v, DppMov<0x112, 0xf>(id, v));         \
  v = OP(v, DppMov<0x114, 0xf>(id, v));         \
  v = OP(v, DppMov<0x118, 0xf>(id, v));         \
  v = OP(v, DppMov<0x142, 0xa>(id, v));         \
  v = OP(v, DppMov<0x143, 0xc>(id, v))
#endif
__device__ __forceinline__ int OpAddI(int a, int b) { return a + b; }
__device__ __forceinline__ int OpMaxI(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int OpMinUBits(int a, int b) { return static_cast<uint32_t>(b) < static_cast<uint32_t>(a) ? b : a; }
__device__ __forceinline__ int OpMinFBits(int a, int b) { return __float_as_int(fminf(__int_as_float(a), __int_as_float(b))); }
// inclusive scans over the 64 lanes
__device__ __forceinline__ int WaveIncSum(int v) {
#ifndef KH_NO_DPP
  KH_DPP_SCAN(v, 0, OpAddI);
#else
  const int lane = KH_TIDX & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(v, o, 64); if (lane >= o) v += n; }
#endif
  return v;
}
__device__ __forceinline__ int WaveIncMax(int v) {   // (values >= 0)
#ifndef KH_NO_DPP
  KH_DPP_SCAN(v, 0, OpMaxI);
#else
  const int lane = KH_TIDX & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(v, o, 64); if (lane >= o) v = OpMaxI(v, n); }
#endif
  return v;
}
__device__ __forceinline__ uint32_t WaveIncMinU(uint32_t u) {
  int v = static_cast<int>(u);
#ifndef KH_NO_DPP
  KH_DPP_SCAN(v, -1, OpMinUBits);
#else
  const int lane = KH_TIDX & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(v, o, 64); if (lane >= o) v = OpMinUBits(v, n); }
#endif
  return static_cast<uint32_t>(v);
}
__device__ __forceinline__ float WaveIncMinF(float f) {
  int v = __float_as_int(f);
#ifndef KH_NO_DPP
  KH_DPP_SCAN(v, 0x7f800000, OpMinFBits);
#else
  const int lane = KH_TIDX & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(v, o, 64); if (lane >= o) v = OpMinFBits(v, n); }
#endif
  return __int_as_float(v);
}
// the value of lane 63 (the total of an inclusive scan), in a scalar register
__device__ __forceinline__ int WaveLast(int v) { return __builtin_amdgcn_readlane(v, 63); }
// value of lane `src` (0 .. 63), per lane: ds_bpermute_b32 directly (__shfl also folds the caller's lane into the
// source index, which costs two more VALU instructions and a live register per call site in a 64-lane wave)
__device__ __forceinline__ int ShflI(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
__device__ __forceinline__ uint32_t ShflU(uint32_t v, int src) { return static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src << 2, static_cast<int>(v))); }
__device__ __forceinline__ float ShflF(float v, int src) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v))); }
__device__ __forceinline__ float WaveMinF(float f) { return __int_as_float(WaveLast(__float_as_int(WaveIncMinF(f)))); }
// 64-bit reductions: the two halves travel separately (identity per half), the total ends in lane 63
#ifndef KH_NO_DPP
template <int kCtrl, int kRowMask>
__device__ __forceinline__ unsigned long long DppMov64(int id_half, unsigned long long v) {
  const uint32_t hi = static_cast<uint32_t>(DppMov<kCtrl, kRowMask>(id_half, static_cast<int>(static_cast<uint32_t>(v >> 32))));
  const uint32_t lo = static_cast<uint32_t>(DppMov<kCtrl, kRowMask>(id_half, static_cast<int>(static_cast<uint32_t>(v))));
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}
#endif
__device__ __forceinline__ unsigned long long OpMinU64(unsigned long long a, unsigned long long b) { return b < a ? b : a; }
__device__ __forceinline__ unsigned long long OpAddU64(unsigned long long a, unsigned long long b) { return a + b; }
__device__ __forceinline__ unsigned long long WaveMinU64ToLast(unsigned long long v) {
#ifndef KH_NO_DPP
  v = OpMinU64(v, DppMov64<0x111, 0xf>(-1, v));
  v = OpMinU64(v, DppMov64<0x112, 0xf>(-1, v));
  v = OpMinU64(v, DppMov64<0x114, 0xf>(-1, v));
  v = OpMinU64(v, DppMov64<0x118, 0xf>(-1, v));
  v = OpMinU64(v, DppMov64<0x142, 0xa>(-1, v));
  v = OpMinU64(v, DppMov64<0x143, 0xc>(-1, v));
#else
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long n = __shfl_xor(v, o, 64); v = n < v ? n : v; }
#endif
  return v;
}
__device__ __forceinline__ long long WaveSumLLToLast(long long x) {
  unsigned long long v = static_cast<unsigned long long>(x);
#ifndef KH_NO_DPP
  v = OpAddU64(v, DppMov64<0x111, 0xf>(0, v));
  v = OpAddU64(v, DppMov64<0x112, 0xf>(0, v));
  v = OpAddU64(v, DppMov64<0x114, 0xf>(0, v));
  v = OpAddU64(v, DppMov64<0x118, 0xf>(0, v));
  v = OpAddU64(v, DppMov64<0x142, 0xa>(0, v));
  v = OpAddU64(v, DppMov64<0x143, 0xc>(0, v));
#else
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
#endif
  return static_cast<long long>(v);
}

// Block primitives with ONE barrier each.  Every primitive writes its per-wave
// partials into a buffer selected by a per-thread call counter (uniform across the
// workgroup) and reads all partials after the barrier; a buffer is rewritten two
// calls later, i.e. behind at least one more barrier than its last read.
template <bool kLdsOnly = false>
__device__ __forceinline__ int BlockExScan(int v, int *total, Blk &sh) {
  const int lane = KH_TIDX & 63, w = KH_TIDX >> 6;
  const int inc = WaveIncSum(v);
  const int buf = NextCall<2>(sh) & 1;
  if (lane == 63) sh->wsum[buf][w] = inc;
  if (kLdsOnly) LdsSync(); else KhSync();
  int before = 0, all = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    const int t = sh->wsum[buf][i];
    before += i < w ? t : 0;
    all += t;
  }
  *total = Uni(all);
  return Uni(before) + inc - v;  // `before` is uniform over the wave
}

// Exclusive scan of K * NT items laid out slice-major (item (k, t) = slice k,
// thread t): one barrier for all slices.
template <int K, bool kLdsOnly = false>
__device__ __forceinline__ void BlockExScanK(const int (&v)[K], int (&off)[K], int *total, Blk &sh) {
  const int lane = KH_TIDX & 63, w = KH_TIDX >> 6;
  const int buf = NextCall<2>(sh) & 1;
  int inc[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    inc[k] = WaveIncSum(v[k]);
    if (lane == 63) sh->wsumk[buf][k][w] = inc[k];
  }
  if (kLdsOnly) LdsSync(); else KhSync();
  int run = 0;
#pragma unroll
  for (int k = 0; k < K; k++) {
    int before = 0, all = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
      const int t = sh->wsumk[buf][k][i];
      before += i < w ? t : 0;
      all += t;
    }
    off[k] = run + Uni(before) + inc[k] - v[k];
    run += Uni(all);
  }
  *total = run;
}

__device__ __forceinline__ unsigned long long BlockMinU64(unsigned long long v, Blk &sh) {
  const int buf = NextCall<1>(sh) & 1;
  v = WaveMinU64ToLast(v);
  if ((KH_TIDX & 63) == 63) sh->wred[buf][KH_TIDX >> 6] = v;
  KhSync();
  unsigned long long r = sh->wred[buf][0];
#pragma unroll
  for (int i = 1; i < NW; i++) r = sh->wred[buf][i] < r ? sh->wred[buf][i] : r;
  return Uni(r);
}

__device__ __forceinline__ float BlockMinF(float v, Blk &sh) {
  const int buf = NextCall<1>(sh) & 1;
  v = WaveIncMinF(v);
  if ((KH_TIDX & 63) == 63) sh->wred[buf][KH_TIDX >> 6] = __float_as_uint(v);
  KhSync();
  float r = __uint_as_float(static_cast<uint32_t>(sh->wred[buf][0]));
#pragma unroll
  for (int i = 1; i < NW; i++) r = fminf(r, __uint_as_float(static_cast<uint32_t>(sh->wred[buf][i])));
  return Uni(r);
}

__device__ __forceinline__ long long BlockSumLL(long long v, Blk &sh) {
  const int buf = NextCall<1>(sh) & 1;
  v = WaveSumLLToLast(v);
  if ((KH_TIDX & 63) == 63) sh->wred[buf][KH_TIDX >> 6] = static_cast<unsigned long long>(v);
  KhSync();
  long long r = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) r += static_cast<long long>(sh->wred[buf][i]);
  return Uni(r);
}

// OR over the workgroup.  Slot k is reset two calls ahead (by thread 0, before the
// barrier of call n it clears the slot of call n + 2): its previous use (call n - 2)
// was fully read before every thread reached the barrier of call n - 1.
template <bool kLdsOnly = false>
__device__ __forceinline__ int BlockOr(int bits, Blk &sh) {
  const int k = NextCall<0>(sh) & 3;
  if (KH_TIDX == 0) sh->orbuf[(k + 2) & 3] = 0;
  if (bits) __hip_atomic_fetch_or(&sh->orbuf[k], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (kLdsOnly) LdsSync(); else KhSync();
  return Uni(sh->orbuf[k]);
}

__device__ __forceinline__ bool BlockAny(bool p, Blk &sh) { return BlockOr(p ? 1 : 0, sh) != 0; }
// (for hand-offs through LDS only: the barrier does not wait for this wave's outstanding stores)
__device__ __forceinline__ bool BlockAnyLds(bool p, Blk &sh) { return BlockOr<true>(p ? 1 : 0, sh) != 0; }

// Exact k-th smallest (0-based) of the cost images tok_cost[b..e): what
// std::nth_element yields at position k (GetCutoff :621-626,:633-640).  Radix select
// over the bits in which the smallest and the largest key of the frame differ (the costs
// of a frame lie within a beam of each other, so their images share the sign, the
// exponent and the top mantissa bits: typically 18-21 bits remain), 11 bits per pass.
constexpr int kRadixBits = 11;
__device__ uint32_t RadixSelect(Arr<const uint32_t> keys, int b, int e, int k, uint32_t kmin, uint32_t kmax,
                                Blk &sh) {
  const uint32_t diff = kmin ^ kmax;
  if (diff == 0) return kmin;
  int remaining = 32 - __clz(static_cast<int>(diff));  // low bits that vary
  uint32_t mask = remaining == 32 ? 0u : ~((1u << remaining) - 1u);
  uint32_t prefix = kmin & mask;
  int passes = (remaining + kRadixBits - 1) / kRadixBits;
  constexpr int kPerLane = ((1 << kRadixBits) + NT - 1) / NT;  // bins a lane owns in the scan
  for (; passes > 0; passes--) {
    const int w = (remaining + passes - 1) / passes, shift = remaining - w;
    const uint32_t dmask = (1u << w) - 1u;
    const int bins = 1 << w;
    for (int i = KH_TIDX; i < bins; i += NT) sh->hist[i] = 0;
    KhSync();
    constexpr int kRU = 8;
    for (int i0 = b + KH_TIDX; i0 < e; i0 += NT * kRU) {
      uint32_t ks[kRU];
#pragma unroll
      for (int k = 0; k < kRU; k++) ks[k] = LoadCostEnc(&keys[min(i0 + k * NT, e - 1)]);
#pragma unroll
      for (int k = 0; k < kRU; k++)
        if (i0 + k * NT < e && (ks[k] & mask) == prefix)
          __hip_atomic_fetch_add(&sh->hist[(ks[k] >> shift) & dmask], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    KhSync();
    // the bin that holds rank k: workgroup scan of the counts (lane t owns bins [t kPerLane, (t + 1) kPerLane))
    int cnt[kPerLane], mine = 0;
#pragma unroll
    for (int j = 0; j < kPerLane; j++) {
      const int bin = KH_TIDX * kPerLane + j;
      cnt[j] = bin < bins ? static_cast<int>(sh->hist[bin]) : 0;
      mine += cnt[j];
    }
    int total;
    int before = BlockExScan(mine, &total, sh);
    if (before <= k && k < before + mine) {
#pragma unroll
      for (int j = 0; j < kPerLane; j++) {
        if (before <= k && k < before + cnt[j]) {
          sh->bcast_i[1] = KH_TIDX * kPerLane + j;
          sh->bcast_i[2] = k - before;
        }
        before += cnt[j];
      }
    }
    KhSync();
    prefix |= static_cast<uint32_t>(Uni(sh->bcast_i[1])) << shift;
    mask |= dmask << shift;
    k = Uni(sh->bcast_i[2]);
    remaining = shift;
  }
  return prefix;
}

// ---------------------------------------------------------------- hash table
__device__ __forceinline__ uint32_t HashState(int32_t s) {
  uint32_t x = static_cast<uint32_t>(s) * 2654435761u;
  return x ^ (x >> 15);
}

// FindOrAddToken (:232-268) without the cost update.  Returns the token index of
// `state` in the frame under construction, creating it if needed (cost slot is
// pre-filled with +inf: arena invariant), or -1 if the raw arena is full.
// Entry: low 32 bits = state + 1 (0 = empty), high 32 bits = token + 1 (0 = pending).
template <bool kQueue>
__device__ int FindOrAdd(const Utt &u, int32_t state, bool has_eps, __attribute__((address_space(3))) int *tok_end /*LDS counter*/,
                         __attribute__((address_space(3))) int *eps_n /*LDS counter*/, int tok_limit, int front_b) {
  uint32_t slot = HashState(state) & u.hash_mask;
  const unsigned long long want_key = static_cast<unsigned long long>(static_cast<uint32_t>(state) + 1u);
  #pragma nounroll
  for (int probes = 0; probes < (1 << 30); probes++) {
    // ONE L2 round trip per probe: the compare-and-swap claims the slot if it is empty
    // and otherwise returns its occupant (a failed CAS is an atomic load)
    unsigned long long ent = kEmpty;
    __hip_atomic_compare_exchange_strong(&u.hash[slot], &ent, want_key, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);  // ent <- previous value
    if (ent == kEmpty) {
      // we own the slot: allocate the token, publish it
      const int idx = __hip_atomic_fetch_add(tok_end, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (idx >= tok_limit) {
        // arena full: publish an invalid token so waiters terminate
        __hip_atomic_exchange(&u.hash[slot], want_key | (0xFFFFFFFFull << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return -1;
      }
      u.tok_state[idx] = state;
      u.tok_extra[idx] = 0.0f;  // "tokens on the currently final frame have zero extra_cost" :241
      u.tmp_slot[idx - front_b] = static_cast<int32_t>(slot);
      if (has_eps) {
        u.tmp_epslist[__hip_atomic_fetch_add(eps_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = idx;
        // kQueue (emitting pass): the list of these tokens IS the first work list of the
        // nonemitting closure — every one of them gets a finite cost from its creator —
        // so it is marked queued here and the emitting pass needs no returning atomics
        if (kQueue) u.tmp_dirty[idx - front_b] = 1;
      }
      // (a no-return L2 swap: measured faster than an 8-byte store through the L1, 0.64 vs 0.71 s for pass 2)
      __hip_atomic_exchange(&u.hash[slot], want_key | (static_cast<unsigned long long>(static_cast<uint32_t>(idx) + 1u) << 32),
                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return idx;
    }
    if ((ent & 0xFFFFFFFFull) == want_key) {
      const uint32_t hi = static_cast<uint32_t>(ent >> 32);
      if (hi == 0xFFFFFFFFu) return -1;
      if (hi != 0) return static_cast<int>(hi - 1u);
      continue;  // pending: the owner publishes within its own loop iteration
    }
    slot = (slot + 1) & u.hash_mask;
  }
  return -1;
}

// Lookup of a state that is known to be in the hash (its token exists).
__device__ __forceinline__ int FindExisting(const Utt &u, int32_t state, unsigned long long first_probe, uint32_t slot) {
  const unsigned long long want_key = static_cast<unsigned long long>(static_cast<uint32_t>(state) + 1u);
  unsigned long long ent = first_probe;
  #pragma nounroll
  for (int probes = 0; probes < (1 << 30); probes++) {
    if ((ent & 0xFFFFFFFFull) == want_key) {
      const uint32_t hi = static_cast<uint32_t>(ent >> 32);
      if (hi != 0) return hi == 0xFFFFFFFFu ? -1 : static_cast<int>(hi - 1u);
    } else {
      slot = (slot + 1) & u.hash_mask;
    }
    ent = __hip_atomic_load(&u.hash[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return -1;
}

// Expansion sweep over the tokens [b, e) (kEps: over entries [b, e) of tmp_epslist, the
// tokens whose state has epsilon arcs): every token whose cost is <= cutoff gets one link
// slot per arc of its HCLG state (arc ranges from the state records `rec`), appended at link slot `lrun` on in token
// order, and body(link slot, token, arc index) runs once per slot, link-parallel.
// Per group of EU * NT tokens: a token sweep (cost, state, arc range), ONE scan that assigns
// the slots, the items' (first slot, first arc, token) through LDS, then one lane per slot
// finds its owner by binary search over the first slots (11 LDS reads) — no seed arrays in
// global memory, no per-token store loops.
// Returns the new end of the link arena, or -1 on overflow (sh->status set).
template <bool kEps, class Body>
__device__ __forceinline__ int ExpandSweep(const Utt &u, Arr<const KhInt4> rec, int b, int e, float cutoff, int lrun,
                                           int frame_cap, long long *arcs, Blk &sh, Body body) {
  const int lrun0 = lrun;
  for (int base = b; base < e; base += NT * EU) {
    int i[EU], st[EU];
    uint32_t co[EU];
    bool in_range[EU];
#pragma unroll
    for (int k = 0; k < EU; k++) {
      i[k] = base + k * NT + KH_TIDX;
      in_range[k] = i[k] < e;
      int ic = min(i[k], e - 1);
      if (kEps) { ic = u.tmp_epslist[ic]; i[k] = ic; }  // [b, e) indexes the list of tokens with epsilon arcs
      co[k] = LoadCostEnc(&u.tok_cost[ic]);
      st[k] = u.tok_state[ic];
      KH_BOUND(1, st[k], 0, 0x7ffffff0);
    }
    int ab[EU], cnt[EU];
#pragma unroll
    for (int k = 0; k < EU; k++) {
      const bool need = in_range[k] && Dec(co[k]) <= cutoff;
      ab[k] = 0;
      cnt[k] = 0;
      if (need) {  // the state's record header: {#emitting arcs (they follow it), first epsilon arc, #epsilon arcs, final}
        const KhInt4 h = rec[st[k]];
        ab[k] = kEps ? h.y : st[k] + 1;
        cnt[k] = kEps ? h.z : h.x;
      }
    }
    int loff[EU], total;
    // (this barrier also orders the previous group's LDS reads before the writes below)
    BlockExScanK<EU>(cnt, loff, &total, sh);
    if (lrun + total > u.link_cap || lrun + total - lrun0 > frame_cap) {
      if (KH_TIDX == 0) sh->status = (lrun + total > u.link_cap) ? 2 : 3;
      KhSync();
      return -1;
    }
#pragma unroll
    for (int k = 0; k < EU; k++) {  // slice-major item order = the scan's order: ex_off is non-decreasing
      sh->ex_off[k * NT + KH_TIDX] = loff[k];
      sh->ex_ab[k * NT + KH_TIDX] = ab[k];
      sh->ex_tok[k * NT + KH_TIDX] = i[k];
    }
    KhSync();
    for (int q = KH_TIDX; q < total; q += NT) {
      // owner = the LAST item whose first slot is <= q (items without arcs share their
      // successor's first slot and are skipped by "last")
      int lo = 0, hi = EU * NT - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (sh->ex_off[mid] <= q) lo = mid; else hi = mid - 1;
      }
      body(lrun + q, sh->ex_tok[lo], sh->ex_ab[lo] + (q - sh->ex_off[lo]));
    }
    if (KH_TIDX == 0) *arcs += total;
    lrun += total;
  }
  KhSync();  // the links are visible to the next phase
  return lrun;
}

// number of set bits of `mask` below this lane: v_mbcnt_lo / v_mbcnt_hi (no per-lane 64-bit mask constant, which the
// sweeps kept in two VGPRs - or reloaded from scratch inside the batch loop)
__device__ __forceinline__ int LanePrefixCount(unsigned long long mask) {
  return static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u)));
}
// One LDS fetch-add per wave, issued by lane 0 alone and returned in a scalar register.  Written with the execution mask
// set by hand: `if (lane == 0) atomic` costs the compare (or two v_readlane for its hoisted, spilled mask), the mask
// save / branch, and on top of that the compiler's atomic optimizer wraps the single-lane atomic in its own
// mbcnt / bcnt / multiply sequence - a dozen instructions per 64-arc batch.  Precondition: lane 0 is active.
__device__ __forceinline__ int WaveLdsFetchAdd(__attribute__((address_space(3))) int *addr, int v) {
#ifndef KH_NO_ASM_ATOMIC
  int r;
  unsigned long long saved;
  asm volatile(
      "s_mov_b64 %1, exec\n\t"
      "s_mov_b64 exec, 1\n\t"
      "ds_add_rtn_u32 %0, %2, %3\n\t"
      "s_mov_b64 exec, %1\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r), "=&s"(saved)
      : "v"(addr), "v"(v)
      : "memory");
  return __builtin_amdgcn_readfirstlane(r);
#else
  int r = 0;
  if ((KH_TIDX & 63) == 0) r = __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __builtin_amdgcn_readfirstlane(r);
#endif
}

// Arc slot -> owning token inside a wave, for the independent-wave sweeps below.  A wave holds 64 tokens with their arc
// counts and first slots (loff, an exclusive prefix sum: non-decreasing); the owner of slot q is the LAST token whose
// first slot is <= q (tokens without arcs share their successor's first slot and are skipped by "last").  Per 64-slot
// batch every token WITH arcs whose first slot lies in the batch writes its lane into that slot of the wave's own LDS
// row (distinct slots: the first slots of tokens with arcs are distinct), and an inclusive maximum scan over the row
// (DPP) hands every slot the last such token at or before it; slot 0 is seeded with the owner that runs on from the
// previous batch.  The entries carry the batch's sequence number above the lane, so a stale entry of an earlier batch
// is smaller than anything written now and the row is cleared once per sweep only.  A wave's LDS instructions execute
// in order: no barrier between the write and the read.  (Rounds 2-4 ran a 6-step binary search over the prefix sums:
// six DEPENDENT ds_bpermute round trips per batch.)
struct OwnerScan {
  __attribute__((address_space(3))) int *row;
  int seq;     // batch sequence number of this wave in units of 128 (lane + 1 sits below it)
  int carry;   // owner of the previous batch's last slot (its arcs may go on in this batch); -1 at the start of a claim
};
__device__ __forceinline__ OwnerScan OwnerScanInit(Blk &sh) {   // (a workgroup barrier must follow before the first batch: none needed, the row is the wave's own)
  OwnerScan os;
  os.row = &sh->own[Uni(static_cast<int>(KH_TIDX >> 6))][0];
  os.row[KH_TIDX & 63] = 0;
  os.seq = 0;
  os.carry = -1;
  return os;
}
__device__ __forceinline__ int OwnerLane(OwnerScan &os, int cnt, int loff, int q0, int lane) {
  os.seq += 128;
  if (cnt > 0 && loff >= q0 && loff < q0 + 64) os.row[loff - q0] = os.seq | (lane + 1);
  __builtin_amdgcn_wave_barrier();
  int ov = os.row[lane];
  if (lane == 0) ov = max(ov, os.seq | (os.carry + 1));
  ov = WaveIncMax(ov);
  const int lo = (ov & 127) - 1;
  os.carry = WaveLast(lo);
  return lo;
}

// ExpandSweep for the emitting arcs with a FILTER: load(token, cost image, arc index) fetches the
// candidate, finish() computes it and says whether it can still be accepted; only those get a
// link slot (store(slot)), appended per wave through one LDS atomic (ballot + prefix count), so
// the slots of a frame are dense.  The reference materialises nothing it rejects either (:731
// "continue"); here a candidate is known to be rejected when it is above an upper bound of the
// frame's final next_cutoff.  Returns the new end of the link arena, or -1 on overflow
// (sh->status set).
//
// INDEPENDENT waves: no workgroup barrier inside the sweep.  A wave
// claims 64 tokens at a time from an LDS cursor, scans their arc counts with shuffles, and maps
// arc slot -> (token, arc) by a 6-step search over the wave's own prefix sums (ds_bpermute; the
// workgroup version searches 2048 entries in LDS behind two barriers per group).  Waves are at
// different stages at any time, so the token / offset / arc round trips of one overlap the
// arithmetic of the others.  The bound is shared through one LDS word (atomic min on the
// order-preserving image): a wave reads it when it claims tokens and lowers it after each
// claim; any stale value is still an upper bound of the final next_cutoff.
template <class Load, class Finish, class Store>
__device__ __forceinline__ int ExpandWavesFiltered(const Utt &u, Arr<const KhInt4> rec, int b, int e, float cutoff, int lrun,
                                                   int frame_cap, long long *arcs, Blk &sh, float *est, float *bound, Load load,
                                                   Finish finish, Store store) {
  const int limit = min(u.link_cap, lrun + frame_cap);
  const int lane = KH_TIDX & 63;
  if (KH_TIDX == 0) {
    sh->link_cursor = lrun;
    sh->work_cursor = b;
    sh->bound_enc = Enc(*bound);
  }
#ifndef KH_OWNER_SEARCH
  OwnerScan os = OwnerScanInit(sh);
#endif
  Arr<uint32_t> w_cost = u.tok_cost;
  Arr<int32_t> w_state = u.tok_state;
  KH_LAUNDER(w_cost.p); KH_LAUNDER(w_state.p);
  KhSync();
  uint32_t my_bound_enc = Enc(*bound);
#ifndef KH_NO_CLAIM_PREFETCH
  // A wave always holds its NEXT claim: cursor and the 64 tokens' cost / state are requested while the current claim's
  // batches run, so a claim starts with the record headers instead of two dependent round trips (554.5 against 557 ms;
  // no change of the kernel's scratch).  KH_CLAIM_PREFETCH2: the next claim's record headers as well (its arc counts),
  // requested from the cost / state that arrived during the claim before - two claims deep.  Measured 559-569 ms
  // against 553: a wave that owns three claims at a time (48 of a frame's ~76 taken up front) balances worse than
  // the round trip is worth; off by default.
  int base_next = WaveLdsFetchAdd(&sh->work_cursor, 64);
  uint32_t co_next = 0u;
  int st_next = 0;
  if (base_next < e) {
    const int ic = min(base_next + lane, e - 1);
    co_next = LoadCostEnc(&w_cost[ic]);
    st_next = w_state[ic];
  }
#ifdef KH_CLAIM_PREFETCH2
  int base_nn = WaveLdsFetchAdd(&sh->work_cursor, 64);
  uint32_t co_nn = 0u;
  int st_nn = 0, cnt_next = 0;
  if (base_nn < e) {
    const int ic = min(base_nn + lane, e - 1);
    co_nn = LoadCostEnc(&w_cost[ic]);
    st_nn = w_state[ic];
  }
  if (base_next < e && base_next + lane < e && Dec(co_next) <= cutoff) cnt_next = rec[st_next].x;
#endif
#endif
  for (;;) {
#if !defined(KH_NO_CLAIM_PREFETCH) && defined(KH_CLAIM_PREFETCH2)
    const int base = base_next;
    if (base >= e) break;
    const int i = base + lane;
    const bool in_range = i < e;
    const uint32_t co = co_next;
    int st = st_next;
    const int cnt_pre = cnt_next;
    // shift the pipeline: claim + 1 becomes the next one (its headers are requested now), a new claim + 2 is taken
    base_next = base_nn; co_next = co_nn; st_next = st_nn;
    cnt_next = 0;
    if (base_next < e && base_next + lane < e && Dec(co_next) <= cutoff) cnt_next = rec[st_next].x;
    base_nn = WaveLdsFetchAdd(&sh->work_cursor, 64);
    if (base_nn < e) {
      const int icn = min(base_nn + lane, e - 1);
      co_nn = LoadCostEnc(&w_cost[icn]);
      st_nn = w_state[icn];
    }
#elif !defined(KH_NO_CLAIM_PREFETCH)
    const int base = base_next;
    if (base >= e) break;
    const int i = base + lane;
    const bool in_range = i < e;
    const uint32_t co = co_next;
    int st = st_next;
    base_next = WaveLdsFetchAdd(&sh->work_cursor, 64);
    if (base_next < e) {
      const int icn = min(base_next + lane, e - 1);
      co_next = LoadCostEnc(&w_cost[icn]);
      st_next = w_state[icn];
    }
#else
    const int base = WaveLdsFetchAdd(&sh->work_cursor, 64);
    if (base >= e) break;
    const int i = base + lane;
    const bool in_range = i < e;
    const int ic = min(i, e - 1);
    const uint32_t co = LoadCostEnc(&w_cost[ic]);
    int st = w_state[ic];
#endif
    KH_BOUND(1, st, 0, 0x7ffffff0);
    const bool need = in_range && Dec(co) <= cutoff;
    int ab = 0, cnt = 0;
    if (need) {  // the emitting arcs follow the state's header in its record
      ab = st + 1;
#if !defined(KH_NO_CLAIM_PREFETCH) && defined(KH_CLAIM_PREFETCH2)
      cnt = cnt_pre;
#else
      cnt = rec[st].x;
#endif
    }
    const int inc = WaveIncSum(cnt);
    const int loff = inc - cnt;
    const int total = WaveLast(inc);
    const float cof = Dec(co);   // (decoded once per token, not once per arc)
    {
      const uint32_t be = Uni(__hip_atomic_load(&sh->bound_enc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      *bound = fminf(*bound, Dec(be));
      my_bound_enc = min(my_bound_enc, be);
    }
#ifndef KH_OWNER_SEARCH
    os.carry = -1;
    const int rel = ab - loff;   // arc index = rel(owner) + slot
#endif
    for (int q0 = 0; q0 < total; q0 += 64) {  // uniform over the wave
      const int q = q0 + lane;
      const bool valid = q < total;
      // owner = the LAST token whose first slot is <= q (tokens without arcs share their
      // successor's first slot and are skipped by "last")
#ifndef KH_OWNER_SEARCH
      const int lo = OwnerLane(os, cnt, loff, q0, lane);
      const int o_rel = ShflI(rel, lo);
      const float o_co = ShflF(cof, lo);
      const int o_ai = o_rel + q;
#else
      int lo = 0, hi = 63;
#pragma unroll
      for (int step = 0; step < 6; step++) {
        const int mid = (lo + hi + 1) >> 1;
        const int v = __shfl(loff, mid, 64);
        if (v <= q) lo = mid; else hi = mid - 1;
      }
      const int o_off = __shfl(loff, lo, 64), o_ab = __shfl(ab, lo, 64);
      const float o_co = __shfl(cof, lo, 64);
      const int o_ai = o_ab + (q - o_off);
#endif
      bool keep = false;
      if (valid) {
        load(0, base + lo, o_co, o_ai);
        keep = finish(0);
      }
      const unsigned long long kb = __ballot(keep);
      if (kb != 0ull) {
        const int n_keep = __popcll(kb);
        const int pos = WaveLdsFetchAdd(&sh->link_cursor, n_keep);
        if (pos + n_keep > limit) {
          if (lane == 0) sh->status = (pos + n_keep > u.link_cap) ? 2 : 3;
        } else if (keep) {
          store(0, pos + LanePrefixCount(kb));
        }
      }
    }
    if (lane == 0) *arcs += total;
    {
      const uint32_t we = Enc(WaveMinF(*est));
      if (we < my_bound_enc) {
        my_bound_enc = we;
        if (lane == 0) __hip_atomic_fetch_min(&sh->bound_enc, we, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  KhSync();  // the links are visible to the next phase
  if (Uni(sh->status) != 0) return -1;
  return Uni(sh->link_cursor);
}

// ---------------------------------------------------------------- frame steps
struct Cutoff {
  float cur_cutoff, adaptive_beam, best_cost;
  int best_tok, count;
  int32_t best_state;   // canonical rule only: the best token's HCLG state (it is part of the reduction key)
};

// GetCutoff :591-658 over the tokens [b, e) of the current frame.  kExact: the best token on a tie is the FIRST one
// in the reference's list order (the strict '<' of :599 / :611), else the one with the smallest state id (rule B).
template <bool kExact = false>
__device__ Cutoff GetCutoff(const Utt &u, const Params &p, int b, int e, Blk &sh) {
  Cutoff c;
  Arr<uint32_t> g_cost = u.tok_cost;
  Arr<int32_t> g_state = u.tok_state;
  KH_LAUNDER_GC(g_cost.p); KH_LAUNDER_GC(g_state.p);
  const int n = e - b;
  c.count = n;
  unsigned long long best = ~0ull;
  int best_i = -1;
  uint32_t kmax = 0;
  // (kGU tokens of a lane in flight together: a plain loop waits for every trip's loads before it issues the next ones)
  constexpr int kGU = 4;
  for (int i0 = b + KH_TIDX; i0 < e; i0 += NT * kGU) {
    uint32_t encs[kGU], lows[kGU];
#pragma unroll
    for (int k = 0; k < kGU; k++) {
      const int i = min(i0 + k * NT, e - 1);
      encs[k] = LoadCostEnc(&g_cost[i]);
      lows[k] = static_cast<uint32_t>(kExact ? UX(x_pos)[i - b] : g_state[i]);
    }
#pragma unroll
    for (int k = 0; k < kGU; k++) {
      const int i = i0 + k * NT;
      if (i >= e) continue;
      // (cost image, state): smallest cost, ties -> smallest state id (canonical rule B)
      const unsigned long long key = (static_cast<unsigned long long>(encs[k]) << 32) | lows[k];
      if (key < best) { best = key; best_i = i; }
      kmax = encs[k] > kmax ? encs[k] : kmax;
    }
  }
  const unsigned long long mine = best;
  best = BlockMinU64(best, sh);
  const float inf = INFINITY;
  if (n == 0) {
    c.best_cost = inf;
    c.best_tok = -1;
    c.best_state = -1;
    c.cur_cutoff = inf;
    c.adaptive_beam = p.beam;
    return c;
  }
  c.best_cost = Dec(static_cast<uint32_t>(best >> 32));
  if (kExact) {
    // the lane that holds the winning key (list positions are unique within a frame) publishes its token
    if (mine == best && best_i >= 0) sh->bcast_i[0] = best_i;
    KhSync();
    c.best_tok = Uni(sh->bcast_i[0]);
    c.best_state = -1;
  } else {
    // the winning key carries the state itself: no broadcast, no barrier, no dependent load of tok_state[best]
    c.best_tok = 0;
    c.best_state = static_cast<int32_t>(static_cast<uint32_t>(best));
  }
  const float best_weight = c.best_cost;
  if (p.max_active == 0x7fffffff && p.min_active == 0) {
    c.adaptive_beam = p.beam;
    c.cur_cutoff = best_weight + p.beam;
    return c;
  }
  const float beam_cutoff = best_weight + p.beam;
  float min_active_cutoff = inf, max_active_cutoff = inf;
  // The k-th smallest costs (nth_element :622-648) only matter when they bind, and one count
  // decides both: with m = #{cost <= beam_cutoff}, the element at sorted index max_active is
  // > beam_cutoff when m <= max_active (the max-active branch is not taken), and the element at
  // sorted index min_active is <= beam_cutoff when m > min_active (the beam branch is taken).
  int within = 0;
  {
    const uint32_t bc = Enc(beam_cutoff);
    constexpr int kWU = 8;
    for (int i0 = b + KH_TIDX; i0 < e; i0 += NT * kWU) {
      uint32_t encs[kWU];
#pragma unroll
      for (int k = 0; k < kWU; k++) encs[k] = LoadCostEnc(&g_cost[min(i0 + k * NT, e - 1)]);
#pragma unroll
      for (int k = 0; k < kWU; k++) within += (i0 + k * NT < e && encs[k] <= bc) ? 1 : 0;
    }
    within = static_cast<int>(BlockSumLL(within, sh));
  }
  // largest cost image of the frame (the smallest is the best cost): bounds the bits the selection looks at
  const uint32_t kmin = static_cast<uint32_t>(best >> 32);
  const bool need_max = n > p.max_active && within > p.max_active;
  const bool need_min = n > p.min_active && p.min_active != 0 && within <= p.min_active;
  if (need_max || need_min) kmax = ~static_cast<uint32_t>(BlockMinU64(static_cast<unsigned long long>(~kmax), sh));
  if (need_max) max_active_cutoff = Dec(RadixSelect(g_cost, b, e, p.max_active, kmin, kmax, sh));
  if (max_active_cutoff < beam_cutoff) {
    c.adaptive_beam = max_active_cutoff - best_weight + p.beam_delta;
    c.cur_cutoff = max_active_cutoff;
    return c;
  }
  if (n > p.min_active) {
    if (p.min_active == 0) min_active_cutoff = best_weight;
    else if (need_min) min_active_cutoff = Dec(RadixSelect(g_cost, b, e, p.min_active, kmin, kmax, sh));
    else min_active_cutoff = beam_cutoff;  // (some value <= beam_cutoff: the beam branch below)
  }
  if (min_active_cutoff > beam_cutoff) {
    c.adaptive_beam = min_active_cutoff - best_weight + p.beam_delta;
    c.cur_cutoff = min_active_cutoff;
  } else {
    c.adaptive_beam = p.beam;
    c.cur_cutoff = beam_cutoff;
  }
  return c;
}

__device__ __forceinline__ float LogLike(const Utt &u, const Params &p, const Blk &sh, int frame, int32_t pdf) {
  if (p.ll_cols > 0) return sh.ll_row[pdf];
  return u.ll[static_cast<size_t>(frame) * u.ll_stride + pdf];
}

// ---------------------------------------------------------------- epsilon closure in LDS
// ProcessNonemitting + the frame's epsilon links without a global atomic.  The closure of an HCLG frame touches a few
// hundred tokens (the benchmark: 330 tokens with epsilon arcs, 32 created by the closure, 328 link slots per frame), but
// the general routine below pays ~8 DEPENDENT L2 / memory round trips per round for them - queued flag (atomic), cost,
// state, record header, arcs, hash compare-and-swap, cost minimum (memory-side atomic), queue flag - 2.5 rounds per
// frame: 12 % of the kernel.  Here the tokens the closure can touch (listed by pass 2: those with epsilon arcs and those
// on states that epsilon arcs lead to) go into an LDS table {state + flags, cost image, token, epsilon arc range}; the
// rounds run on the table - one read-only gather of the arcs per round is the only memory access - with work lists and
// queued flags in LDS; new tokens get their arena index from the LDS counter as before; when the fixed point is reached
// the costs and the new tokens are written back with plain stores and the links are made from the table.  The fixed
// point is the least one, whatever the schedule: costs, token set and links equal the general routine's.  A frame whose
// list does not fit (or a table that fills up on the way: nothing has been written to memory by then) takes the general
// routine.  The global hash is not used by such a frame at all: no ClearHash, no hash slot per token.
constexpr int kClSlots = 2048, kClMaxLoad = 1300, kClList = 2048;
constexpr uint32_t kClQueued = 0x80000000u, kClUnknown = 0xFFFFFFFFu;
typedef __attribute__((address_space(3))) uint32_t *LdsU32;
typedef __attribute__((address_space(3))) uint16_t *LdsU16;
struct ClTab {
  LdsU32 key, cost, idx, rng;   // [kClSlots]: state + flags + 1 (0 = empty) | Enc(cost) | token - frontier begin (+ queued bit) | epsilon arcs: first << 7 | count (127 = read the header)
  LdsU16 l0, l1;                // [kClList] work lists (slots), double buffered
};
static_assert(4 * kClSlots <= kLdsSlots && 2 * kClList * sizeof(uint16_t) <= sizeof(unsigned int) * (1 << 11), "closure table over the token table's values, work lists over the histogram");
__device__ __forceinline__ ClTab ClLayout(const Blk &sh) {
  ClTab t;
  LdsU32 v = (LdsU32)LdsVals(sh);
  t.key = v; t.cost = v + kClSlots; t.idx = v + 2 * kClSlots; t.rng = v + 3 * kClSlots;
  LdsU16 h = (LdsU16)LdsKeys(sh);
  t.l0 = h; t.l1 = h + kClList;
  return t;
}
// slot of `key` (inserted if absent: *inserted); -1 = the table is full
__device__ __forceinline__ int ClFindOrAdd(const ClTab &t, uint32_t key, bool *inserted) {
  uint32_t slot = HashState(static_cast<int32_t>(key)) & (kClSlots - 1);
#pragma nounroll
  for (int probes = 0; probes < kClSlots; probes++) {
    uint32_t seen = 0u;
    __hip_atomic_compare_exchange_strong(&t.key[slot], &seen, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (seen == 0u) { *inserted = true; return static_cast<int>(slot); }
    if (seen == key) { *inserted = false; return static_cast<int>(slot); }
    slot = (slot + 1) & (kClSlots - 1);
  }
  return -1;
}
__device__ __forceinline__ int ClFind(const ClTab &t, uint32_t key) {
  uint32_t slot = HashState(static_cast<int32_t>(key)) & (kClSlots - 1);
#pragma nounroll
  for (int probes = 0; probes < kClSlots; probes++) {
    const uint32_t seen = t.key[slot];
    if (seen == key) return static_cast<int>(slot);
    if (seen == 0u) return -1;
    slot = (slot + 1) & (kClSlots - 1);
  }
  return -1;
}
__device__ __forceinline__ uint32_t ClPackRange(int ab, int cnt) {
  return (static_cast<uint32_t>(ab) < (1u << 25)) ? ((static_cast<uint32_t>(ab) << 7) | static_cast<uint32_t>(cnt < 127 ? cnt : 127)) : 127u;
}

// Returns 1 = done (closure + links of frame `frame`), 0 = not applicable (nothing changed: take the general routine),
// -1 = failed (sh->status set).
__device__ int ClosureLds(const Utt &u, const Params &p, int frame, float cutoff, Blk &sh) {
  const int n_list = Uni(sh->erel_n);
  if (n_list > p.cl_max_load || p.cl_max_load == 0) return 0;
  const ClTab t = ClLayout(sh);
  const int fb = Uni(sh->front_b), tok_end0 = Uni(sh->tok_end);
  const int tok_limit = min(u.tok_cap, fb + u.tok_frame_cap);
  for (int i = KH_TIDX; i < kClSlots; i += NT) { t.key[i] = 0u; t.cost[i] = kEncInf; t.idx[i] = 0u; }   // (rng: set by whoever inserts)
  if (KH_TIDX == 0) { sh->cl_n = n_list; sh->flag = 0; sh->wl_n[0] = 0; sh->wl_n[1] = 0; sh->wl_n[2] = 0; }
  LdsSync();
  // ---- the table from the list; the tokens with epsilon arcs are the first work list (:766-767)
  for (int k = KH_TIDX; k < n_list; k += NT) {
    const int32_t ns = u.tmp_work0[k];
    const int idx = u.tmp_work1[k];
    const uint32_t enc = __float_as_uint(u.tmp_f0[k]);
    bool ins;
    const int slot = ClFindOrAdd(t, static_cast<uint32_t>(ns) + 1u, &ins);   // (states are unique within a frame; <= kClMaxLoad entries: never full)
    t.cost[slot] = enc;
    uint32_t iv = static_cast<uint32_t>(idx - fb);
    t.rng[slot] = kClUnknown;
    if ((ns & kHasEps) != 0) {
      const KhInt4 h = p.rec[ns & kStateMask];
      t.rng[slot] = ClPackRange(h.y, h.z);
      iv |= kClQueued;
      t.l0[__hip_atomic_fetch_add(&sh->wl_n[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = static_cast<uint16_t>(slot);
    }
    t.idx[slot] = iv;
  }
  LdsSync();
  // ---- rounds
  long long my_arcs = 0;
  for (int r = 0;; r++) {
    const int n = Uni(sh->wl_n[r % 3]);
    if (KH_TIDX == 0) sh->wl_n[(r + 2) % 3] = 0;
    if (n == 0) break;
    const LdsU16 cur = (r & 1) ? t.l1 : t.l0, nxt = (r & 1) ? t.l0 : t.l1;
    auto nxt_n = &sh->wl_n[(r + 1) % 3];
    for (int q = KH_TIDX; q < n; q += NT) {
      const int s = cur[q];
      // leave the queue BEFORE reading the cost (a later improvement queues the token again; the LDS unit executes the
      // workgroup's operations one after the other, so "clear, then read" cannot miss an update that found the flag set)
      (void)__hip_atomic_fetch_and(&t.idx[s], ~kClQueued, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const float cur_cost = Dec(__hip_atomic_load(&t.cost[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (cur_cost > cutoff) continue;  // :779
      uint32_t rg = t.rng[s];
      int ab, cnt;
      if (rg == kClUnknown || (rg & 127u) == 127u) {
        const KhInt4 h = p.rec[static_cast<int32_t>(t.key[s] - 1u) & kStateMask];
        ab = h.y;
        cnt = h.z;
        if (rg == kClUnknown) t.rng[s] = ClPackRange(ab, cnt);
      } else {
        ab = static_cast<int>(rg >> 7);
        cnt = static_cast<int>(rg & 127u);
      }
      for (int j = 0; j < cnt; j++) {
        const KhInt4 arc = p.n_arcs[ab + j];
        my_arcs++;
        const float tot_cost = cur_cost + __int_as_float(arc.z);
        if (!(tot_cost < cutoff)) continue;  // :794
        bool ins;
        const int d = ClFindOrAdd(t, static_cast<uint32_t>(arc.w & (kStateMask | kHasEps | kEpsDst)) + 1u, &ins);
        if (d < 0) { sh->flag = 1; continue; }
        if (ins) {
          const int cn = __hip_atomic_fetch_add(&sh->cl_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (cn >= p.cl_max_load) sh->flag = 1;
          const int idx = __hip_atomic_fetch_add(&sh->tok_end, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (idx >= tok_limit) sh->status = 1;
          t.rng[d] = kClUnknown;   // (read by the token's own processing only, a barrier from here)
          (void)__hip_atomic_fetch_or(&t.idx[d], static_cast<uint32_t>(idx - fb) & ~kClQueued, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const uint32_t enc = Enc(tot_cost);
        const uint32_t old = __hip_atomic_fetch_min(&t.cost[d], enc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (enc < old && (arc.w & kHasEps) != 0) {   // "changed": new or cheaper -> (re)process the destination
          const uint32_t was = __hip_atomic_fetch_or(&t.idx[d], kClQueued, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if ((was & kClQueued) == 0u) {
            const int pos = __hip_atomic_fetch_add(nxt_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (pos < kClList) nxt[pos] = static_cast<uint16_t>(d); else sh->flag = 1;
          }
        }
      }
    }
    if (u.phase_cycles != nullptr && KH_TIDX == 0) sh->phase[13] += 1;
    LdsSync();
    if (Uni(sh->flag) != 0 && Uni(sh->status) == 0) {
      // the table (or a work list) filled up: nothing has reached memory - undo the token reservations and let the
      // general routine do the frame
      LdsSync();
      if (KH_TIDX == 0) { sh->tok_end = tok_end0; sh->flag = 0; sh->wl_n[0] = sh->eps_n; sh->wl_n[1] = 0; sh->wl_n[2] = 0; }
      LdsSync();
      return 0;
    }
    if (Uni(sh->status) != 0) return -1;
  }
  LdsSync();
  // ---- write back: final costs; the tokens the closure created (those with epsilon arcs join tmp_epslist)
  for (int s = KH_TIDX; s < kClSlots; s += NT) {
    const uint32_t key = t.key[s];
    if (key == 0u) continue;
    const int32_t ns = static_cast<int32_t>(key - 1u);
    const int i = fb + static_cast<int>(t.idx[s] & ~kClQueued);
    u.tok_cost[i] = t.cost[s];
    if (i >= tok_end0) {
      u.tok_state[i] = ns & kStateMask;
      u.tok_extra[i] = 0.0f;  // :241
      if ((ns & kHasEps) != 0) u.tmp_epslist[__hip_atomic_fetch_add(&sh->eps_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = i;
    }
  }
  Stamp(u, sh, 3);
  // ---- epsilon links {(tok, arc): cost[tok] <= cutoff}: one slot per epsilon arc of every token under the cutoff, a
  // token's slots consecutive and in arc order; the slots whose tot_cost is not under the cutoff stay dead (dst = -1)
  const int blk_b = Uni(sh->link_end);
  const int limit = min(u.link_cap, blk_b + u.link_frame_cap);
  if (KH_TIDX == 0) sh->link_cursor = blk_b;
  LdsSync();
  for (int s0 = 0; s0 < kClSlots; s0 += NT) {   // (uniform: the wave scan involves every lane)
    const int s = s0 + KH_TIDX;
    const uint32_t key = t.key[s];
    const float cost = Dec(t.cost[s]);
    int ab = 0, cnt = 0;
    if (key != 0u && (static_cast<int32_t>(key - 1u) & kHasEps) != 0 && !(cost > cutoff)) {
      const uint32_t rg = t.rng[s];
      if (rg == kClUnknown || (rg & 127u) == 127u) {
        const KhInt4 h = p.rec[static_cast<int32_t>(key - 1u) & kStateMask];
        ab = h.y;
        cnt = h.z;
      } else {
        ab = static_cast<int>(rg >> 7);
        cnt = static_cast<int>(rg & 127u);
      }
    }
    const int inc = WaveIncSum(cnt);
    const int total = WaveLast(inc);
    if (total == 0) continue;   // (uniform over the wave)
    const int base = WaveLdsFetchAdd(&sh->link_cursor, total);
    if (base + total > limit) {
      if ((KH_TIDX & 63) == 0) sh->status = (base + total > u.link_cap) ? 2 : 3;
      continue;
    }
    const int src = fb + static_cast<int>(t.idx[s] & ~kClQueued);
    int l = base + inc - cnt;
    for (int j = 0; j < cnt; j++, l++) {
      const int ai = ab + j;
      const KhInt4 arc = p.n_arcs[ai];
      const float g = __int_as_float(arc.z), tot_cost = cost + g;
      int dst = -1;
      float k = 0.0f;
      if (tot_cost < cutoff) {  // the token exists
        const int d = ClFind(t, static_cast<uint32_t>(arc.w & (kStateMask | kHasEps | kEpsDst)) + 1u);
        if (d >= 0) {
          dst = fb + static_cast<int>(t.idx[d] & ~kClQueued);
          // the constant part of link_extra_cost (:309-311), (cost[src] + 0 + g) - cost[dst]: both costs are final
          k = (cost + 0.0f + g) - Dec(t.cost[d]);
        }
      }
      u.link_dst[l] = dst;
      u.link_src[l] = src;
      u.link_arc[l] = -1 - ai;
      u.link_k[l] = k;
    }
  }
#ifndef KH_CL_BLOCKSUM
  {  // the arcs visited, for the utterance's counter: one LDS add per wave (no barrier of its own)
    const long long wave_arcs = WaveSumLLToLast(my_arcs);
    if ((KH_TIDX & 63) == 63 && wave_arcs != 0)
      __hip_atomic_fetch_add(&sh->arcs_expanded, wave_arcs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  KhSync();
  if (Uni(sh->status) != 0) return -1;
  if (KH_TIDX == 0) {
    sh->link_end = sh->link_cursor;
    u.feps_b[frame] = blk_b;
    u.feps_e[frame] = sh->link_end;
  }
#else
  KhSync();
  if (Uni(sh->status) != 0) return -1;
  const long long tot_arcs = BlockSumLL(my_arcs, sh);
  if (KH_TIDX == 0) {
    sh->link_end = sh->link_cursor;
    sh->arcs_expanded += tot_arcs;
    u.feps_b[frame] = blk_b;
    u.feps_e[frame] = sh->link_end;
  }
#endif
  KhSync();
  Stamp(u, sh, 4);
  return 1;
}

// A frame that does not take ClosureLds: what pass 2 no longer does for the general routine - the listed tokens on
// epsilon-destination states enter the global hash, every frontier token gets its hash slot (-1 = none) for ClearHash,
// and the tokens with epsilon arcs are marked queued.
__device__ void ClosureGeneralPrep(const Utt &u, Blk &sh) {
  const int fb = Uni(sh->front_b), fe = Uni(sh->tok_end), n_list = Uni(sh->erel_n);
  for (int i = fb + KH_TIDX; i < fe; i += NT) u.tmp_slot[i - fb] = -1;
  if (KH_TIDX == 0) sh->hash_dirty = 1;
  KhSync();
  for (int k = KH_TIDX; k < n_list; k += NT) {
    const int32_t ns = u.tmp_work0[k];
    const int idx = u.tmp_work1[k];
    if ((ns & kHasEps) != 0) u.tmp_dirty[idx - fb] = 1;
    if ((ns & kEpsDst) != 0) {
      const unsigned long long want = static_cast<unsigned long long>(static_cast<uint32_t>(ns & kStateMask) + 1u) |
                                      (static_cast<unsigned long long>(static_cast<uint32_t>(idx) + 1u) << 32);
      uint32_t g = HashState(ns & kStateMask) & u.hash_mask;
#pragma nounroll
      for (int probes = 0; probes < (1 << 30); probes++) {
        unsigned long long ent = kEmpty;
        __hip_atomic_compare_exchange_strong(&u.hash[g], &ent, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ent == kEmpty) break;
        g = (g + 1) & u.hash_mask;
      }
      u.tmp_slot[idx - fb] = static_cast<int32_t>(g);
    }
  }
  KhSync();
}

// ProcessNonemitting :752-812 on the tokens of the frame under construction
// ([sh->front_b, sh->tok_end)), then generation of the epsilon links with the
// converged costs.  Returns false on arena overflow.
__device__ bool ProcessNonemitting(const Utt &u_in, const Params &p_in, int frame, float cutoff, Blk &sh, bool from_list = true) {
  // the phase's own copies of the pointers it uses (see ProcessEmitting: kept in scalar registers for its duration)
  Utt u = u_in;
  Params p = p_in;
  KH_LAUNDER_NE(u.tmp_epslist.p); KH_LAUNDER_NE(u.tmp_work0.p); KH_LAUNDER_NE(u.tmp_work1.p); KH_LAUNDER_NE(u.tmp_dirty.p);
  KH_LAUNDER_NE(u.tok_cost.p); KH_LAUNDER_NE(u.tok_state.p); KH_LAUNDER_NE(u.tok_extra.p); KH_LAUNDER_NE(u.tmp_slot.p);
  KH_LAUNDER_NE(u.hash.p); KH_LAUNDER_NE(u.hash_mask);
  KH_LAUNDER_NE(u.link_dst.p); KH_LAUNDER_NE(u.link_src.p); KH_LAUNDER_NE(u.link_arc.p); KH_LAUNDER_NE(u.link_k.p);
  KH_LAUNDER_NE(p.rec.p); KH_LAUNDER_NE(p.n_arcs.p);
  KH_LAUNDER_NE(u.feps_b.p); KH_LAUNDER_NE(u.feps_e.p); KH_LAUNDER_NE(u.tmp_f0.p);
  if (from_list) {   // (false: frame 0, whose start token DecodeInit entered in the global hash itself)
#ifndef KH_NO_LDS_CLOSURE
    const int rc = ClosureLds(u, p, frame, cutoff, sh);
    if (u.phase_cycles != nullptr && KH_TIDX == 0) sh->phase[rc != 0 ? 38 : 39] += 1;
    if (rc != 0) return rc > 0;
#endif
    ClosureGeneralPrep(u, sh);
  }
  const int fb = Uni(sh->front_b);
  const int tok_limit = min(u.tok_cap, fb + u.tok_frame_cap);
  // ---- cost fixed point: min-plus closure under the cutoff, driven by work lists.
  // List 0 = the first wl_n[0] entries of tmp_epslist: the tokens with epsilon arcs
  // that the emitting pass created (marked queued at creation); processing a token may
  // lower the cost of others, which are queued (once: tmp_dirty) for the next round.
  long long my_arcs = 0;
  for (int r = 0;; r++) {
    const int n = Uni(sh->wl_n[r % 3]);
    if (KH_TIDX == 0) sh->wl_n[(r + 2) % 3] = 0;  // last read one barrier ago, next pushed to one barrier ahead
    if (n == 0) break;
    const Arr<const int32_t> cur = r == 0 ? u.tmp_epslist : ((r & 1) ? u.tmp_work1 : u.tmp_work0);
    const Arr<int32_t> nxt = (r & 1) ? u.tmp_work0 : u.tmp_work1;
    auto nxt_n = &sh->wl_n[(r + 1) % 3];
    const int lane = KH_TIDX & 63;
    long long tcl = 0;
    const bool prof = u.phase_cycles != nullptr && KH_TIDX == 0;
    if (prof) tcl = static_cast<long long>(__builtin_amdgcn_s_memtime());
#define KH_CL_STAMP(k) do { if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const long long now_ = static_cast<long long>(__builtin_amdgcn_s_memtime()); sh->phase[k] += now_ - tcl; tcl = now_; } } while (0)
    if (prof) sh->phase[33] += n;
    for (int q0 = 0; q0 < n; q0 += NT) {  // uniform trip count: the wave-level combine below involves every lane
      const int q = q0 + KH_TIDX;
      float cur_cost = INFINITY;
      int ab = 0, ae = 0;
      if (q < n) {
        const int i = cur[q];
        // leave the queue BEFORE reading the cost: a later improvement queues the token again.
        // The flag and the cost are different words (different L2 channels): the store must have
        // been performed before the load is issued, or an improver could see the stale flag (no
        // re-queue) while this lane still reads the old cost - a lost update.
        (void)__hip_atomic_exchange(&u.tmp_dirty[i - fb], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cur_cost = Dec(LoadCostEnc(&u.tok_cost[i]));
        if (!(cur_cost > cutoff)) {  // :779
          const int32_t s = u.tok_state[i];
          const KhInt4 h = p.rec[s];
          ab = h.y;
          ae = h.y + h.z;
        }
      }
      KH_CL_STAMP(27);
      // The arcs of the wave's tokens in lockstep.  In an HCLG the epsilon arcs of a frame
      // lead to a handful of states (word ends -> language-model states -> the back-off
      // state): hundreds of lanes doing FindOrAdd + atomicMin on ONE hash slot / cost word
      // serialise in L2 (35 % of the kernel on the structured workload).  The lanes of a wave
      // that target the same state are combined first - their minimum goes out once, by the
      // first of them; all leaders then issue their global operations together.
      for (int j = 0; __ballot(ab + j < ae) != 0ull; j++) {
        bool ok = false;
        int32_t ds = 0;
        uint32_t enc = 0xFFFFFFFFu;
        if (ab + j < ae) {
          const KhInt4 arc = p.n_arcs[ab + j];
          my_arcs++;
          const float graph_cost = __int_as_float(arc.z), tot_cost = cur_cost + graph_cost;
          if (tot_cost < cutoff) {  // :794
            ok = true;
            ds = arc.w;
            enc = Enc(tot_cost);
          }
        }
        unsigned long long todo = __ballot(ok);
        bool lead = ok;
        uint32_t gmin = enc;
        for (int it = 0; todo != 0ull && it < 16; it++) {  // (past 16 distinct states the rest go out one by one)
          const int leader = __ffsll(static_cast<long long>(todo)) - 1;
          const int32_t key = __builtin_amdgcn_readlane(ds, leader);
          const bool mine = ok && ds == key;
          const unsigned long long grp = __ballot(mine);
          const uint32_t m = static_cast<uint32_t>(WaveLast(static_cast<int>(WaveIncMinU(mine ? enc : 0xFFFFFFFFu))));
          if (mine) {
            gmin = m;
            lead = lane == leader;
          }
          todo &= ~grp;
        }
        KH_CL_STAMP(28);
        if (ok && lead) {
          const bool he = (ds & kHasEps) != 0;
          const int dst = FindOrAdd<false>(u, ds & kStateMask, he, &sh->tok_end, &sh->eps_n, tok_limit, fb);
          if (dst < 0) { sh->status = 1; continue; }
          if (he) {
            const uint32_t old = __hip_atomic_fetch_min(&u.tok_cost[dst], gmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gmin < old &&  // "changed": new or cheaper -> (re)process dst
                __hip_atomic_exchange(&u.tmp_dirty[dst - fb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
              nxt[__hip_atomic_fetch_add(nxt_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = dst;
          } else {
            (void)__hip_atomic_fetch_min(&u.tok_cost[dst], gmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        KH_CL_STAMP(29);
      }
    }
    if (u.phase_cycles != nullptr && KH_TIDX == 0) sh->phase[13] += 1;
    KhSync();
    KH_CL_STAMP(30);
#undef KH_CL_STAMP
    if (Uni(sh->status) != 0) return false;
  }
  KhSync();
  Stamp(u, sh, 3);
  // ---- epsilon links: {(tok, arc): cost[tok] <= cutoff, cost[tok] + w < cutoff}
  // One slot per epsilon arc of every token under the cutoff; the slots whose
  // tot_cost is not under the cutoff stay dead (dst = -1) until the compaction.
  const int fe = Uni(sh->tok_end);
  const int blk_b = Uni(sh->link_end);
  long long seeded = 0;
  const int blk_e = ExpandSweep<true>(
      u, p.rec, 0, Uni(sh->eps_n), cutoff, blk_b, u.link_frame_cap, &seeded, sh, [&](int l, int src, int ai) {
        KH_BOUND(3, src, 0, u.tok_cap);
        KH_BOUND(4, ai, 0, p.num_eps);
        const KhInt4 arc = p.n_arcs[ai];
        const uint32_t co = LoadCostEnc(&u.tok_cost[src]);
        const float g = __int_as_float(arc.z), tot_cost = Dec(co) + g;
        int dst = -1;
        if (tot_cost < cutoff) {  // the token exists already
          const uint32_t slot = HashState(arc.w & kStateMask) & u.hash_mask;
          const unsigned long long ent = __hip_atomic_load(&u.hash[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          dst = FindExisting(u, arc.w & kStateMask, ent, slot);
        }
        u.link_dst[l] = dst;
        u.link_src[l] = src;
        u.link_arc[l] = -1 - ai;
        // the constant part of link_extra_cost (:309-311), (cost[src] + 0 + g) - cost[dst]: both costs
        // are final once the closure has converged (the acoustic cost of an epsilon link is 0: not stored)
        u.link_k[l] = dst >= 0 ? (Dec(co) + 0.0f + g) - Dec(LoadCostEnc(&u.tok_cost[dst])) : 0.0f;
      });
  if (blk_e < 0) return false;
  KhSync();
  if (KH_TIDX == 0) sh->link_end = blk_e;
  const long long tot_arcs = BlockSumLL(my_arcs, sh);
  if (KH_TIDX == 0) {
    sh->arcs_expanded += tot_arcs;
    u.feps_b[frame] = blk_b;
    u.feps_e[frame] = sh->link_end;
  }
  KhSync();
  Stamp(u, sh, 4);
  return true;
}

// Clears the hash entries of the frontier tokens [fb, fe) (inserted this frame).
__device__ void ClearHash(const Utt &u, int fb, int fe, Blk &sh) {
  if (Uni(sh->hash_dirty) == 0) return;   // (the frame's closure ran in LDS: the global hash was not touched)
  if (2 * (fe - fb) > static_cast<int>(u.hash_mask >> 2)) {
    // many entries: stream over the whole table (coalesced) instead of one scattered store per token
    for (uint32_t i = KH_TIDX; i <= u.hash_mask; i += NT) u.hash[i] = kEmpty;
  } else {
    for (int i = fb + KH_TIDX; i < fe; i += NT) {
      const int32_t sl = u.tmp_slot[i - fb];  // -1: the token was never entered (emitting pass, no epsilon arc leads to its state)
      if (sl >= 0) u.hash[sl] = kEmpty;
    }
  }
  KhSync();
  if (KH_TIDX == 0) sh->hash_dirty = 0;
}

// ---- pass 2 of ProcessEmitting (shared by the canonical and the reference-order sweep): accept `tot_cost <= next_cutoff`
// (the canonical rule E tests the frame's FINAL next_cutoff here; the reference-order sweep has applied the running
// cutoff already and passes +inf) and FindOrAddToken +
// cost minimum IN LDS.  Every global atomic of this pool executes at the memory side (one
// DRAM read-modify-write per lane: TCC_EA0_ATOMIC == TCC_ATOMIC), and the CAS + publish +
// min per accepted arc were 85 % of the kernel's atomics.  The accepted candidates are
// split by a hash of their state into P parts of <= ~2800 candidates; each part is deduped
// in the 4096-slot LDS table (CAS on the key, min on the cost image), the occupied slots get
// consecutive token indices from one scan (a deterministic order), the tokens are written
// with plain stores and a second sweep gives the part's links their token index.  Only the
// tokens the epsilon closure may look up (kEpsDst states) also enter the global hash.
// kLocal + fuse_ord (reference order): phase (D) also gives every new token its INSERTION KEY - the smallest ordinal among
// its candidates (x_ord, from the acceptance sweep) - and the caller's id of its state (x_csid): the token's table value
// holds (smallest ordinal so far << 13 | the token's index within the part) from (C) on, a candidate folds its ordinal in
// with one LDS minimum, the first one to arrive stores the state id, and a sweep over the table's slots writes the keys out
// in token order.  (Until round 5 a separate sweep over the candidates did this after pass 2: 8 % of the kernel.)
template <bool kLocal>
__device__ __forceinline__ bool EmitPass2(const Utt &u, Blk &sh, int nb, int tok_limit, int link_frame_b, int link_frame_e,
                                          float next_cutoff, int n_acc_known = 0, bool fuse_ord = false) {
  static_assert(kLdsSlots % NT == 0, "slots per lane");
  // the pass's own copies of the pointers it uses (see ProcessEmitting) - in the reference-order kernel only:
  // same-box A/B, canonical kernel 691 ms with them / 676 ms without, reference-order kernel 2208 / 2246 ms
  Arr<float> e_k = u.link_k, e_extra = u.tok_extra;
  Arr<int32_t> e_dst = u.link_dst, e_state = u.tok_state, e_epslist = u.tmp_epslist, e_work0 = u.tmp_work0, e_work1 = u.tmp_work1;
  Arr<uint32_t> e_cost = u.tok_cost;
  Arr<float> e_f0 = u.tmp_f0;
  if constexpr (kLocal) {
    KH_LAUNDER_P2(e_f0.p);
    KH_LAUNDER_P2(e_k.p); KH_LAUNDER_P2(e_extra.p); KH_LAUNDER_P2(e_dst.p); KH_LAUNDER_P2(e_state.p); KH_LAUNDER_P2(e_epslist.p);
    KH_LAUNDER_P2(e_work0.p); KH_LAUNDER_P2(e_work1.p); KH_LAUNDER_P2(e_cost.p);
  }
  auto keys = LdsKeys(sh);
  auto vals = LdsVals(sh);
  const float nan = __int_as_float(0x7fc00000);
  auto part_of = [](uint32_t h, int parts) { return static_cast<int>(((h >> 12) * static_cast<uint32_t>(parts)) >> 20); };
  // LOCALITY: the table is hashed by BLOCKS of 2^kLocBits consecutive state ids (h = hash of the
  // block), a block's states take consecutive slots, and the slot scan of (C) numbers the tokens
  // in slot order — so the tokens of neighbouring states are neighbours in the next frame's
  // expansion: their arc-offset words share a cache line and their arcs are contiguous in the
  // arc table (an HCLG numbers the states of an HMM chain / a lexicon-tree branch consecutively).
#ifndef KH_PART_CAND
#define KH_PART_CAND 11000   // accepted candidates per part of pass 2 (7 k / 9 k / 13 k / 15 k measured: +6 % / 0 / +1 % / +8 %)
#endif
#ifndef KH_LOC_BITS
#define KH_LOC_BITS 5
#endif
  // A state id is the unit index of its record (header + emitting arcs: ~3 units for the states of
  // an HMM chain), so a block of 2^kLocBits table slots stands for 2^(kLocBits + kLocShift) units.
#ifndef KH_LOC_SHIFT
#define KH_LOC_SHIFT 1
#endif
  constexpr int kLocBits = KH_LOC_BITS, kLocShift = KH_LOC_SHIFT;
  auto lds_slot = [](uint32_t h, int32_t ns) {
    return ((h << kLocBits) | ((static_cast<uint32_t>(ns) >> kLocShift) & ((1u << kLocBits) - 1u))) & (kLdsSlots - 1);
  };
  // number of parts: about 11 000 accepted candidates per part (typically half as many
  // distinct states: a load of ~0.65); a part whose table fills up is redone with twice the
  // parts (the parts nest, and resolved links are marked, so nothing is done twice)
  int parts = 1;
  if constexpr (kLocal) {   // (reference order: the caller has counted the live candidates)
    while (parts * KH_PART_CAND < n_acc_known) parts *= 2;
  } else if (link_frame_e - link_frame_b > KH_PART_CAND) {
    int n_acc_mine = 0;
    {
      constexpr int kCU = 8;   // loads of a lane in flight together (a frame of this size took 11+ dependent round trips here)
      for (int l0 = link_frame_b + KH_TIDX; l0 < link_frame_e; l0 += NT * kCU) {
        float ks[kCU];
#pragma unroll
        for (int k = 0; k < kCU; k++) ks[k] = e_k[min(l0 + k * NT, link_frame_e - 1)];
#pragma unroll
        for (int k = 0; k < kCU; k++) n_acc_mine += (l0 + k * NT < link_frame_e && !(ks[k] > next_cutoff)) ? 1 : 0;
      }
    }
    const int n_acc = static_cast<int>(BlockSumLL(n_acc_mine, sh));
    while (parts * KH_PART_CAND < n_acc) parts *= 2;
    if (u.phase_cycles != nullptr && KH_TIDX == 0) sh->phase[32] += n_acc;
  }
  if (u.phase_cycles != nullptr && KH_TIDX == 0) { sh->phase[31] += link_frame_e - link_frame_b; sh->phase[13] += 0; }
  for (int k = 0; k < parts; k++) {
    for (int i = KH_TIDX; i < kLdsSlots; i += NT) { keys[i] = 0u; vals[i] = 0xFFFFFFFFu; }
    if (KH_TIDX == 0) sh->flag = 0;
    KhSync();
    KM(40, k);
    // (B) insert.  A link that an earlier part resolved holds its token index (>= 0), a rejected one -1,
    // an unresolved one -2 - (next state + flags).  kMU
    // candidates per lane are loaded before any is used (independent loads in flight).
    constexpr int kMU = 4;
    // a lane's kMU = 4 consecutive candidates: one 16-byte load per array (element by element, clamped, in the frame's last group)
    bool last_read = false;   // (D) of the last part is the last read of the candidates' cost and state before pruning
    bool dst_only = false;    // (D) in reference order: the costs are not looked at again (every live candidate has been accepted; NaNs never got here)
    auto load_group = [&](int base, float (&tc)[kMU], int32_t (&nsv)[kMU]) {
      if (kLocal && dst_only && base + kMU <= link_frame_e) {
        const KhInt4 n4 = last_read ? Load4I_NT(e_dst, base) : Load4I(e_dst, base);
        tc[0] = 0.0f; tc[1] = 0.0f; tc[2] = 0.0f; tc[3] = 0.0f;
        nsv[0] = n4.x; nsv[1] = n4.y; nsv[2] = n4.z; nsv[3] = n4.w;
      } else if (base + kMU <= link_frame_e) {
#ifndef KH_NO_NT
        const KhFloat4 t4 = last_read ? Load4F_NT(e_k, base) : Load4F(e_k, base);
        const KhInt4 n4 = last_read ? Load4I_NT(e_dst, base) : Load4I(e_dst, base);
#else
        const KhFloat4 t4 = Load4F(e_k, base);
        const KhInt4 n4 = Load4I(e_dst, base);
#endif
        tc[0] = t4.x; tc[1] = t4.y; tc[2] = t4.z; tc[3] = t4.w;
        nsv[0] = n4.x; nsv[1] = n4.y; nsv[2] = n4.z; nsv[3] = n4.w;
      } else {
#pragma unroll
        for (int j = 0; j < kMU; j++) {
          const int l = base + j;
          const int lc = l < link_frame_e ? l : link_frame_e - 1;
          tc[j] = e_k[lc];
          nsv[j] = e_dst[lc];
          if (l >= link_frame_e) tc[j] = nan;
        }
      }
    };
    // KH_P2_PREFETCH: the NEXT group's loads issued before this group's candidates are inserted (the sweep waits for a
    // round trip per trip: loads, s_waitcnt, work - ten in a row per part for a 40 k-candidate frame, twice per part).
    // Measured: 575-581 ms against 558 without - the eight registers of the group in flight push the kernel's scratch
    // from 144 to 164 bytes per lane, which costs more than the round trips; off by default.
    float tc[kMU], tc_next[kMU];
    int32_t nsv[kMU], nsv_next[kMU];
    {
      const int base0 = link_frame_b + KH_TIDX * kMU;
      if (base0 < link_frame_e) load_group(base0, tc, nsv);
    }
    for (int base = link_frame_b + KH_TIDX * kMU; base < link_frame_e; base += NT * kMU) {
#ifdef KH_P2_PREFETCH
      const bool more = base + NT * kMU < link_frame_e;
#else
      const bool more = false;
      if (base != link_frame_b + static_cast<int>(KH_TIDX) * kMU) load_group(base, tc, nsv);
#endif
      if (more) load_group(base + NT * kMU, tc_next, nsv_next);
#pragma unroll
      for (int j = 0; j < kMU; j++) {
        const float tot_cost = tc[j];
        if (nsv[j] >= -1 || tot_cost > next_cutoff || tot_cost != tot_cost) continue;  // :731 "if (tot_cost > next_cutoff) continue"
        const int32_t ns = -2 - nsv[j];
        const uint32_t h = HashState((ns & kStateMask) >> (kLocBits + kLocShift));
        if (part_of(h, parts) != k) continue;
        const uint32_t key = static_cast<uint32_t>(ns) + 1u;
        uint32_t slot = lds_slot(h, ns);
        // double hashing BY BLOCK: a block whose place is taken moves as a whole (same offset within
        // the block, a step that depends on the block only), so its tokens stay neighbours; linear
        // probing piles the runs of consecutive active states up (pass 2 twice as slow).
        // (Round 4, thread-0 stamps inside this pass: count sweep 6 %, clear + insert 45 %, slot scan + token writes 27 %,
        // resolve sweep 22 % of its cycles.  The lane's four probe sequences side by side - every round issues the
        // compare-and-swaps of all candidates still looking for a slot before it waits - measured +1.5 % on the
        // kernel: the insert is not bound by the chain of LDS round trips.)
        const uint32_t step = ((h >> 9) | 1u) << kLocBits;
        int probes = 0;
        // (not unrolled: the compiler's 8-fold unrolling of this early-exit loop cost ~100 scalar mask instructions per
        // candidate; most candidates stop at the first or second probe)
#pragma nounroll
        for (; probes < 256; probes++) {
          uint32_t seen = 0u;
          __hip_atomic_compare_exchange_strong(&keys[slot], &seen, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (seen == 0u || seen == key) break;
          slot = (slot + step) & (kLdsSlots - 1);
        }
        if (probes == 256) { sh->flag = 1; continue; }  // the table is (nearly) full
        (void)__hip_atomic_fetch_min(&vals[slot], Enc(tot_cost), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (more) {
#pragma unroll
        for (int j = 0; j < kMU; j++) { tc[j] = tc_next[j]; nsv[j] = nsv_next[j]; }
      }
    }
    KhSync();
    if (Uni(sh->flag) != 0) {  // redo from this part on with twice as many parts (nothing was written yet)
      KhSync();                // (every lane has read the flag before it is reset)
      if (parts >= (1 << 20)) {
        if (KH_TIDX == 0) sh->status = 5;
        KhSync();
        return false;
      }
      parts *= 2;
      k = 2 * k - 1;
      continue;
    }
    KM(41, k);
    // (C) occupied slots -> consecutive token indices; tokens written with plain stores
    const int tok_base = Uni(sh->tok_end);
    int occ[kLdsSlots / NT], off[kLdsSlots / NT], total;
#pragma unroll
    for (int j = 0; j < kLdsSlots / NT; j++) occ[j] = keys[KH_TIDX + j * NT] != 0u ? 1 : 0;
    BlockExScanK<kLdsSlots / NT>(occ, off, &total, sh);
    if (tok_base + total > tok_limit) {
      if (KH_TIDX == 0) sh->status = 1;
      KhSync();
      return false;
    }
#pragma unroll
    for (int j = 0; j < kLdsSlots / NT; j++) {
      if (!occ[j]) continue;
      const int i = KH_TIDX + j * NT;
      const int32_t ns = static_cast<int32_t>(keys[i] - 1u);
      const int idx = tok_base + off[j];
      e_state[idx] = ns & kStateMask;
      const uint32_t cost_enc = vals[i];
      e_cost[idx] = cost_enc;
#ifndef KH_NO_NT
      __builtin_nontemporal_store(0.0f, &e_extra[idx]);  // "tokens on the currently final frame have zero extra_cost" :241
#else
      e_extra[idx] = 0.0f;  // "tokens on the currently final frame have zero extra_cost" :241
#endif
      if (kLocal && fuse_ord) vals[i] = (0x7FFFFu << 13) | static_cast<uint32_t>(off[j]);
      else vals[i] = static_cast<uint32_t>(idx);
      // these tokens are the closure's first work list (every one has a finite cost)
      if ((ns & kHasEps) != 0) {
        const int je = __hip_atomic_fetch_add(&sh->eps_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        e_epslist[je] = idx;
        if constexpr (kLocal) UX(x_c0e)[je] = cost_enc;   // (reference order: the closure's replay starts from these costs)
      }
      // The tokens the epsilon closure can touch - those with epsilon arcs and those whose state is the destination of
      // one - are LISTED (state + flags, token); the closure builds its LDS table from the list (ClosureLds), or, for the
      // rare frame that does not fit it, enters them in the global hash first (ClosureGeneralPrep).  Rounds 2-4 entered the
      // kEpsDst tokens in the global hash here: a memory-side compare-and-swap per token on the pass's critical path,
      // plus the hash slot and the queued flag of EVERY new token written for ClearHash / the work lists.
      if ((ns & (kHasEps | kEpsDst)) != 0) {
        const int k = __hip_atomic_fetch_add(&sh->erel_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        e_work0[k] = ns;
        e_work1[k] = idx;
        e_f0[k] = __uint_as_float(cost_enc);   // (the cost image, bit for bit: saves the closure a dependent gather)
      }
    }
    if (KH_TIDX == 0) sh->tok_end = tok_base + total;
    KhSync();
    KM(42, k);
    // (D) the part's links get their token index; rejected candidates become dead links
    last_read = k + 1 == parts;
    dst_only = kLocal;
    {
      const int base0 = link_frame_b + KH_TIDX * kMU;
      if (base0 < link_frame_e) load_group(base0, tc, nsv);
    }
    for (int base = link_frame_b + KH_TIDX * kMU; base < link_frame_e; base += NT * kMU) {
      const bool full = base + kMU <= link_frame_e;
#ifdef KH_P2_PREFETCH
      const bool more = base + NT * kMU < link_frame_e;
#else
      const bool more = false;
      if (base != link_frame_b + static_cast<int>(KH_TIDX) * kMU) load_group(base, tc, nsv);
#endif
      if (more) load_group(base + NT * kMU, tc_next, nsv_next);   // (as in (B): the next group's loads under this group's work)
      // reference order: the group's ordinals and state ids requested WITH its costs and destinations (one 16-byte load
      // per array) - read where they are used, inside the loop below, each was a dependent round trip behind the stores of
      // the candidate before it (the compiler cannot move a load over a store that may alias): round 6's fine stamps
      // had this pass at 166 k cycles per frame against the canonical kernel's 100 k
      uint32_t ordv[kMU] = {0u, 0u, 0u, 0u};
      int32_t csidv[kMU] = {0, 0, 0, 0};
      if (kLocal && fuse_ord) {
        if (full) {
          const KhInt4 o4 = Load4I_NT(UX(x_ord), base - link_frame_b), c4 = Load4I_NT(UX(x_csid), base - link_frame_b);
          ordv[0] = static_cast<uint32_t>(o4.x); ordv[1] = static_cast<uint32_t>(o4.y); ordv[2] = static_cast<uint32_t>(o4.z); ordv[3] = static_cast<uint32_t>(o4.w);
          csidv[0] = c4.x; csidv[1] = c4.y; csidv[2] = c4.z; csidv[3] = c4.w;
        } else {
#pragma unroll
          for (int j = 0; j < kMU; j++) {
            const int lc = min(base + j, link_frame_e - 1) - link_frame_b;
            ordv[j] = static_cast<uint32_t>(UX(x_ord)[lc]);
            csidv[j] = UX(x_csid)[lc];
          }
        }
      }
      bool wrote = false;
#pragma unroll
      for (int j = 0; j < kMU; j++) {
        const int l = base + j;
        if (l >= link_frame_e || nsv[j] >= -1) continue;   // resolved by an earlier part, or rejected
        const float tot_cost = tc[j];
        if (tot_cost != tot_cost || tot_cost > next_cutoff) {
          nsv[j] = -1;  // rejected (:731; a NaN candidate too)
          wrote = true;
          if (!full) e_dst[l] = -1;
          continue;
        }
        const int32_t ns = -2 - nsv[j];
        const uint32_t h = HashState((ns & kStateMask) >> (kLocBits + kLocShift));
        if (part_of(h, parts) != k) continue;
        const uint32_t key = static_cast<uint32_t>(ns) + 1u;
        uint32_t slot = lds_slot(h, ns);
        const uint32_t step = ((h >> 9) | 1u) << kLocBits;
#pragma nounroll
        while (keys[slot] != key) slot = (slot + step) & (kLdsSlots - 1);
        if (kLocal && fuse_ord) {
          const uint32_t ord = ordv[j];
          const uint32_t lidx = vals[slot] & 8191u;
          const uint32_t old = __hip_atomic_fetch_min(&vals[slot], (ord << 13) | lidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          nsv[j] = tok_base + static_cast<int32_t>(lidx);
          if ((old >> 13) == 0x7FFFFu) UX(x_bkt)[nsv[j] - nb] = csidv[j];   // (the first candidate of the token to get here)
        } else {
          nsv[j] = static_cast<int32_t>(vals[slot]);
        }
        wrote = true;
        if (!full) e_dst[l] = nsv[j];
      }
      if (full && wrote) {   // the lane owns the four slots: one 16-byte store
        KhInt4 o4;
        o4.x = nsv[0]; o4.y = nsv[1]; o4.z = nsv[2]; o4.w = nsv[3];
#ifndef KH_NO_NT
        if (k + 1 == parts) Store4I_NT(e_dst, base, o4); else Store4I(e_dst, base, o4);   // (the last part: nobody reads the links before pruning)
#else
        Store4I(e_dst, base, o4);
#endif
      }
      if (more) {
#pragma unroll
        for (int j = 0; j < kMU; j++) { tc[j] = tc_next[j]; nsv[j] = nsv_next[j]; }
      }
    }
    KhSync();
    if (kLocal && fuse_ord) {   // the part's insertion keys, in token order
#pragma unroll
      for (int j = 0; j < kLdsSlots / NT; j++) {
        const int i = KH_TIDX + j * NT;
        if (keys[i] == 0u) continue;
        const uint32_t v = vals[i];
        UX(x_q)[tok_base + static_cast<int>(v & 8191u) - nb] = v >> 13;
      }
      KhSync();
    }
  }
  return true;
}

// The frame's score row -> LDS with all of a lane's loads in flight together.  (`for (c = t; c < n; c += NT) lds[c] = g[c]`
// compiles to one load, one s_waitcnt vmcnt(0), one store per trip: six DEPENDENT HBM round trips for a 5800-column row
// at the head of every frame.  hipcc does not software-pipeline a loop whose trip count it does not know.)
__device__ __forceinline__ void StageScoreRow(const Utt &u, const Params &p, Blk &sh, int frame) {
  constexpr int kU = 8;
  GP(const float) src = u.ll + static_cast<size_t>(frame) * u.ll_stride;
  for (int c0 = KH_TIDX; c0 < p.ll_cols; c0 += NT * kU) {
    float v[kU];
#pragma unroll
#ifndef KH_NO_NT
    for (int k = 0; k < kU; k++) v[k] = c0 + k * NT < p.ll_cols ? __builtin_nontemporal_load(&src[c0 + k * NT]) : 0.0f;
#else
    for (int k = 0; k < kU; k++) v[k] = c0 + k * NT < p.ll_cols ? src[c0 + k * NT] : 0.0f;
#endif
#pragma unroll
    for (int k = 0; k < kU; k++)
      if (c0 + k * NT < p.ll_cols) sh.ll_row[c0 + k * NT] = v[k];
  }
}

// ProcessEmitting :660-750 for frame `frame` (tokens [b, e) -> new tokens appended
// at sh->tok_end, which becomes the new frontier sh->front_b).  Returns next_cutoff
// through *next_cutoff_out; false on overflow.
__device__ bool ProcessEmitting(const Utt &u, const Params &p, int frame, int b, int e,
                                float *next_cutoff_out, int *cand_out, Blk &sh) {
  const int nb = Uni(sh->tok_end);  // first token of frame + 1
  const int tok_limit = min(u.tok_cap, nb + u.tok_frame_cap);
  if (KH_TIDX == 0) { sh->wl_n[0] = 0; sh->wl_n[1] = 0; sh->eps_n = 0; sh->erel_n = 0; }  // pass 2 fills tmp_epslist (barriers in between)
  // stage the frame's acoustic scores in LDS (the barriers of GetCutoff order it
  // against the last readers of the previous row and the first readers of this one)
  StageScoreRow(u, p, sh, frame);
  Stamp(u, sh, 15);
  const Cutoff c = GetCutoff(u, p, b, e, sh);
  Stamp(u, sh, 0);
  if (KH_TIDX == 0 && c.count > sh->max_tokens_frame) sh->max_tokens_frame = c.count;
  const float inf = INFINITY;
  float cost_offset = 0.0f;
  float est = inf;
  if (c.best_tok >= 0) {
    cost_offset = -c.best_cost;  // :691
    // :692-704 estimate from the best token's arcs (different association order
    // from the main loop: ((w + (offset - ll)) + tot_cost) + adaptive_beam)
    const int32_t s = c.best_state;
    const float tot = c.best_cost;
    // The record's header (arc count) and its first 64 arcs are requested TOGETHER - the first wave reads the units
    // behind the header before it knows how many of them are arcs of this state (they are units of the same table;
    // the index is clamped to it) - one round trip instead of two dependent ones in front of the frame's expansion.
    const int ab = s + 1;
    const int a0 = ab + KH_TIDX;
    KhInt4 arc0;
    arc0.x = 0; arc0.y = 0; arc0.z = 0; arc0.w = 0;
    if (KH_TIDX < 64) arc0 = p.rec[min(a0, p.num_units - 1)];
    const int ae = ab + p.rec[s].x;
    if (KH_TIDX < 64 && a0 < ae) {
      const float w = __int_as_float(arc0.z) + (cost_offset - LogLike(u, p, sh, frame, arc0.x));   // {pdf, olabel, weight, nextstate}
      const float new_weight = w + tot;
      est = fminf(est, new_weight + c.adaptive_beam);
    }
    for (int a = a0 + (KH_TIDX < 64 ? NT : 0); a < ae; a += NT) {
      const KhInt4 arc = p.rec[a];
      const float w = __int_as_float(arc.z) + (cost_offset - LogLike(u, p, sh, frame, arc.x));
      const float new_weight = w + tot;
      est = fminf(est, new_weight + c.adaptive_beam);
    }
  }
  if (KH_TIDX == 0) u.cost_offset[frame] = cost_offset;  // :710-711

  // ---- pass 1: the emitting arcs of every token under cur_cutoff (token sweep + scan, then
  // one lane per arc): fetches the arc and the acoustic score, reduces
  // min(tot_cost + adaptive_beam) and materialises the candidates that can still be accepted.
  // est0 = the estimate from the best token's arcs (:692-704) is an upper bound of the final
  // next_cutoff, so a candidate above it is rejected whatever the rest of the frame holds.
  const float est0 = BlockMinF(est, sh);
  est = est0;
  float bound = est0;  // upper bound of the final next_cutoff; tightens as the sweep proceeds
  const int link_frame_b = Uni(sh->link_end);
  long long my_arcs = 0;
  constexpr int kLU = 1;  // (2 in flight per lane: no gain, +16 B of scratch per lane)
  KhInt4 c_arc[kLU];
  int c_src[kLU], c_ai[kLU];
  float c_co[kLU];
  float c_ac[kLU], c_tot[kLU];
  // The sweep's own copies of the pointers it uses (a scalar move each, once per frame): short-lived values with all
  // their uses inside the sweep, which the register allocator keeps in scalar registers for its duration - the
  // originals live for the whole launch and were reloaded from their spill lanes in front of every access of the
  // 64-arc batch loop (22 v_readlane per batch in the round-4 listing).
  Arr<const KhInt4> x_rec = p.rec;
  Arr<int32_t> x_dst = u.link_dst, x_src = u.link_src, x_arc = u.link_arc;
  Arr<float> x_k = u.link_k, x_a = u.link_a;
  GP(const float) x_ll = u.ll + static_cast<size_t>(frame) * u.ll_stride;
  int x_keep_ac = p.keep_ac, x_ll_cols = p.ll_cols;
  KH_LAUNDER(x_rec.p); KH_LAUNDER(x_dst.p); KH_LAUNDER(x_src.p); KH_LAUNDER(x_arc.p); KH_LAUNDER(x_k.p); KH_LAUNDER(x_a.p);
  KH_LAUNDER(x_ll); KH_LAUNDER(x_keep_ac); KH_LAUNDER(x_ll_cols);
  const int link_frame_e = ExpandWavesFiltered(
      u, x_rec, b, e, c.cur_cutoff, link_frame_b, u.link_frame_cap, &my_arcs, sh, &est, &bound,
      [&](int k, int src, float src_cost, int ai) {
        KH_BOUND(5, src, 0, u.tok_cap);
        KH_BOUND(6, ai, 0, p.num_units);
        c_arc[k] = x_rec[ai];
        c_ai[k] = ai;
        c_src[k] = src;
        c_co[k] = src_cost;
      },
      [&](int k) -> bool {
        int32_t pdf = c_arc[k].x;
        KH_BOUND(7, pdf, 0, u.ll_stride);
        const float like = x_ll_cols > 0 ? sh.ll_row[pdf] : x_ll[pdf];
        c_ac[k] = cost_offset - like;
        c_tot[k] = c_co[k] + c_ac[k] + __int_as_float(c_arc[k].z);  // :726-730
        est = fminf(est, c_tot[k] + c.adaptive_beam);
        return !(c_tot[k] > bound);
      },
      [&](int k, int l) {
        x_dst[l] = -2 - c_arc[k].w;  // <= -2: the HCLG next state (+ flags), unresolved; token index after pass 2
#ifndef KH_NO_NT
        // (source, arc and acoustic cost of a link are not read again before the next pruning visit: non-temporal stores -
        // same-box A/B 546 -> 535 ms, with the other nt accesses of the file 531: the lines they no longer displace in L2
        // are arc records the next frames re-read)
        __builtin_nontemporal_store(c_src[k], &x_src[l]);
        __builtin_nontemporal_store(c_ai[k], &x_arc[l]);
        if (x_keep_ac) __builtin_nontemporal_store(c_ac[k], &x_a[l]);
#else
        x_src[l] = c_src[k];
        x_arc[l] = c_ai[k];
        if (x_keep_ac) x_a[l] = c_ac[k];
#endif
        x_k[l] = c_tot[k];
      });
  if (link_frame_e < 0) return false;
  // final next_cutoff: the value the reference's running cutoff converges to
  const float next_cutoff = BlockMinF(est, sh);
  Stamp(u, sh, 1);
  if (KH_TIDX == 0) {
    u.femit_b[frame] = link_frame_b;
    u.femit_e[frame] = link_frame_e;
    sh->link_end = link_frame_e;
    sh->front_b = nb;
  }
  KhSync();

  if (!EmitPass2<false>(u, sh, nb, tok_limit, link_frame_b, link_frame_e, next_cutoff)) return false;
  KhSync();
  const long long tot_arcs = BlockSumLL(my_arcs, sh);
  if (KH_TIDX == 0) {
    sh->arcs_expanded += tot_arcs;
    sh->wl_n[0] = sh->eps_n;  // the closure's first work list = the new tokens with epsilon arcs
  }
  KhSync();
  Stamp(u, sh, 2);
  *next_cutoff_out = next_cutoff;
  *cand_out = link_frame_e - link_frame_b;
  return Uni(sh->status) == 0;
}

// ================================================================ exact reference order
// Params::exact_order (kh_decoder_set_reference_order): the search of LatticeFasterDecoder with the reference's OWN
// iteration order, so that the token and link sets equal the reference's bit for bit (oracle mode 0), not only the
// order-independent resolution (mode 3).  What depends on the order (DESIGN.md "Decoder parity"):
//   (1) ProcessEmitting accepts an arc iff tot_cost <= the RUNNING next_cutoff (:728-733), i.e. the minimum of the
//       estimate from the best token (:688-705) and of tot_cost + adaptive_beam over the arcs visited BEFORE it, tokens in
//       the order of the HashList's list (hash-list-inl.h:118-147), arcs in arc order;
//   (2) GetCutoff's best token is the first minimum in that order (:599, :611);
//   (3) the list order of the next frame = buckets (state % hash_size) in the order of their first occupation, elements
//       of a bucket in insertion order; insertions = the accepted arcs in visiting order, then the insertions of
//       ProcessNonemitting's LIFO queue (:766-811) in the order that queue makes them; hash_size follows :219-225.
// How it is computed here:
//   * every frontier token carries its list position (x_pos).  A first sweep over the emitting arcs reduces, per token,
//     min(tot_cost) (x_m, by list position) and the arc count (x_c); ONE scan in list order turns them into the running
//     cutoff in front of the token and the ordinal of its first arc; the second sweep accepts arc k of a token against
//     min(that, the token's earlier arcs) - a segmented prefix minimum inside the wave - and materialises only what the
//     reference would, each candidate with its ordinal;
//   * the LDS token table of pass 2 keeps, per new state, the minimum cost AND the minimum ordinal = its insertion key;
//   * the epsilon closure runs as the parallel fixed point it always was (costs, token set and links do not depend on the
//     order), and the ORDER of its insertions is then replayed by one lane over the frame's epsilon links only (a few
//     hundred pops per frame on an HCLG), from LDS;
//   * bucket minima (one atomicMin per token into a table of hash_size words), the key (first occupation of the bucket,
//     insertion key) and one stable radix sort give the next frame's list positions.
// The backward pruning needs no order: the lattice FinalizeDecoding returns does not depend on the sweep order or on
// delta (Params::lazy_prune explains why; tests hold oracle mode 0 against mode 2 on thousands of random cases).
// hash_size starts at 1000 for every utterance (a freshly constructed decoder, lattice-faster-decoder.cc:37).

// Workgroup exclusive scan of (sum, min) pairs: ONE barrier.
__device__ __forceinline__ void BlockExScanSumMin(int v, uint32_t m, int *ex_sum, uint32_t *ex_min, int *tot_sum, uint32_t *tot_min, Blk &sh) {
  const int lane = KH_TIDX & 63, w = KH_TIDX >> 6;
  const int inc = WaveIncSum(v);
  const uint32_t im = WaveIncMinU(m);
  const int b1 = NextCall<2>(sh) & 1, b2 = NextCall<1>(sh) & 1;
  if (lane == 63) {
    sh->wsum[b1][w] = inc;
    sh->wred[b2][w] = im;
  }
#ifndef KH_NO_DPP
  const uint32_t pm = static_cast<uint32_t>(DppMov<0x138, 0xf>(-1, static_cast<int>(im)));   // wave_shr:1; lane 0 keeps the identity
#else
  uint32_t pm = static_cast<uint32_t>(__shfl_up(static_cast<int>(im), 1, 64));
  if (lane == 0) pm = 0xFFFFFFFFu;
#endif
  KhSync();
  int before = 0, all = 0;
  uint32_t bm = 0xFFFFFFFFu, am = 0xFFFFFFFFu;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    const int t = sh->wsum[b1][i];
    const uint32_t tm = static_cast<uint32_t>(sh->wred[b2][i]);
    if (i < w) {
      before += t;
      bm = tm < bm ? tm : bm;
    }
    all += t;
    am = tm < am ? tm : am;
  }
  bm = Uni(bm);
  *ex_sum = Uni(before) + inc - v;
  *ex_min = pm < bm ? pm : bm;
  *tot_sum = Uni(all);
  *tot_min = Uni(am);
}

// Stable LSD radix sort of the n (key, value) pairs in (x_key0, x_val0) by the low `bits` bits of each 32-bit half of the
// key (the other bits are zero by construction), 9 bits per pass.  Per pass: per-wave digit histograms in LDS (over the
// idle token-table area), one scan in (digit, wave) order, and a scatter in which the lanes of a tile that hold the same
// digit find each other with ballots (rank = lanes before me with my digit).  Waves own contiguous segments and walk
// them in order, so the sort is stable.  Returns 0 / 1: the buffer pair that holds the result.
constexpr int kSortBits = 9;
static_assert(NW * (1 << kSortBits) <= kLdsSlots, "per-wave digit histograms fit the LDS token-table area");
__device__ int BlockRadixSort(const Utt &u, int n, int bits, Blk &sh) {
  auto hist = LdsKeys(sh);   // [NW][1 << kSortBits]
  constexpr int kBins = 1 << kSortBits;
  const int w = KH_TIDX >> 6, lane = KH_TIDX & 63;
  const int seg = ((n + NW - 1) / NW + 63) & ~63;
  const int s0 = min(n, w * seg), s1 = min(n, s0 + seg);
  int cur = 0;
  for (int half = 0; half < 2; half++) {
    for (int sb = 0; sb < bits; sb += kSortBits) {
      const int shift = half * 32 + sb;
      const Arr<const unsigned long long> kin = cur ? UX(x_key1) : UX(x_key0);
      const Arr<const int32_t> vin = cur ? UX(x_val1) : UX(x_val0);
      const Arr<unsigned long long> kout = cur ? UX(x_key0) : UX(x_key1);
      const Arr<int32_t> vout = cur ? UX(x_val0) : UX(x_val1);
      for (int i = KH_TIDX; i < NW * kBins; i += NT) hist[i] = 0u;
      LdsSync();
      constexpr int kTU = 4;   // tiles of a wave's segment whose loads are in flight together (the passes are latency-bound)
      for (int base = s0; base < s1; base += 64 * kTU) {
        unsigned long long k4[kTU];
#pragma unroll
        for (int j = 0; j < kTU; j++) {
          const int i = base + j * 64 + lane;
          k4[j] = i < s1 ? kin[i] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < kTU; j++) {
          if (base + j * 64 + lane < s1)
            __hip_atomic_fetch_add(&hist[w * kBins + (static_cast<uint32_t>(k4[j] >> shift) & (kBins - 1))], 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      LdsSync();
      {  // exclusive scan in (digit, wave) order: entry idx = digit * NW + wave; a lane owns NW * kBins / NT consecutive entries
        constexpr int kPer = NW * kBins / NT;
        static_assert(kPer * NT == NW * kBins, "entries per lane");
        uint32_t c[kPer];
        int mine = 0;
#pragma unroll
        for (int j = 0; j < kPer; j++) {
          const int idx = KH_TIDX * kPer + j;
          c[j] = hist[(idx % NW) * kBins + idx / NW];
          mine += static_cast<int>(c[j]);
        }
        int total;
        int run = BlockExScan<true>(mine, &total, sh);
#pragma unroll
        for (int j = 0; j < kPer; j++) {
          const int idx = KH_TIDX * kPer + j;
          hist[(idx % NW) * kBins + idx / NW] = static_cast<uint32_t>(run);
          run += static_cast<int>(c[j]);
        }
      }
      LdsSync();
      for (int base = s0; base < s1; base += 64 * kTU) {   // (uniform over the wave)
        unsigned long long k4[kTU];
        int32_t v4[kTU];
#pragma unroll
        for (int j = 0; j < kTU; j++) {
          const int i = base + j * 64 + lane;
          k4[j] = i < s1 ? kin[i] : 0ull;
          v4[j] = i < s1 ? vin[i] : 0;
        }
#pragma unroll
        for (int j = 0; j < kTU; j++) {
          const bool act = base + j * 64 + lane < s1;
          const uint32_t d = static_cast<uint32_t>(k4[j] >> shift) & (kBins - 1);
          unsigned long long peers = __ballot(act);
          if (peers == 0ull) continue;
#pragma unroll
          for (int bit = 0; bit < kSortBits; bit++) {
            const bool set = ((d >> bit) & 1u) != 0u;
            const unsigned long long m = __ballot(set);
            peers &= set ? m : ~m;
          }
          if (act) {
            const int rank = LanePrefixCount(peers);
            const uint32_t at = hist[w * kBins + d];
            if (rank == 0) hist[w * kBins + d] = at + static_cast<uint32_t>(__popcll(peers));
            kout[at + rank] = k4[j];
            vout[at + rank] = v4[j];
          }
        }
      }
      KhSync();   // the scattered pairs are the next pass's input
      cur ^= 1;
    }
  }
  return cur;
}

// The LIFO queue of ProcessNonemitting (:766-811) replayed by ONE lane over the frame's epsilon links: which insertion
// comes when.  ncost: cost of the tokens with epsilon arcs (index = entry of tmp_epslist; +inf = not inserted yet), nl0 /
// nl1: their link slots, lcode: destination of a link slot (-1 dead or without effect on the order; an entry of
// tmp_epslist; 0x40000000 | (token - nb) for a token without epsilon arcs that the closure creates), lw: arc weight.
// Only tokens with epsilon arcs are pushed (popping another one is a no-op in the reference).  Returns the number of
// insertions, or -1 when the queue outgrew its arrays.
template <class FP, class IP>
__device__ int ReplayClosure(FP ncost, IP nl0, IP nl1, IP lcode, FP lw, IP stack_lo, int lo_cap, const Utt &u, const Blk &sh, int top,
                             float cutoff, int nb, uint32_t qbase, int n_new) {
  int cnt = 0;
  const int hi_cap = u.link_frame_cap;
  // (once every token the closure creates has been inserted the rest of the queue only moves costs, which are known)
  while (top > 0 && cnt < n_new) {
    --top;
    const int uu = top < lo_cap ? stack_lo[top] : UX(x_stack)[top - lo_cap];
    const float c = ncost[uu];
    if (c > cutoff) continue;   // :779
    const int l1 = nl1[uu];
    for (int l = nl0[uu]; l < l1; l++) {
      const int code = lcode[l];
      if (code < 0) continue;
      const float tot = c + lw[l];
      if (!(tot < cutoff)) continue;   // :794
      if ((code & 0x40000000) != 0) {
        const int k = code & 0x3fffffff;
        if (UX(x_q)[k] == 0xFFFFFFFFu) UX(x_q)[k] = qbase + static_cast<uint32_t>(cnt++);   // FindOrAddToken inserts it; changed, pushed, popped without effect
      } else {
        const float cv = ncost[code];
        bool push = false;
        if (cv == INFINITY) {   // new: inserted, changed
          UX(x_q)[u.tmp_epslist[code] - nb] = qbase + static_cast<uint32_t>(cnt++);
          push = true;
        } else if (tot < cv) {  // :252-254 cheaper: changed
          push = true;
        }
        if (push) {
          ncost[code] = tot;
          if (top < lo_cap) stack_lo[top] = code;
          else if (top - lo_cap < hi_cap) UX(x_stack)[top - lo_cap] = code;
          else return -1;
          top++;
        }
      }
    }
  }
  return cnt;
}

// The same replay by ONE WAVE with all its state in LDS.  Most pops change nothing (a word-end token whose epsilon arc
// leads to a language-model state that an earlier pop has already given a cheaper cost): the wave takes the next 64 queue
// entries, every lane evaluates its entry against the CURRENT costs, and only the first lane whose pop would insert or
// improve something is executed (by lane 0, depth first with an LDS stack, exactly as the queue would run it); the lanes
// before it are no-ops - costs only fall, and nothing has changed since they were evaluated - and the lanes after it are
// evaluated again.  LDS link codes: -1 dead; 0x40000000 | (token - ne_emit) a token without epsilon arcs the closure
// creates; else entry of tmp_epslist in bits 0-11 and, if the closure creates that token, (token - ne_emit) + 1 in bits 12-.
// queue: the initial queue in list order (popped from its end).  Returns the insertions, -1 if the depth-first stack
// outgrew its LDS array (the caller then runs ReplayClosure from memory).
constexpr int kRpNodes = kLdsSlots / 4, kRpLinks = kLdsSlots / 2;
typedef __attribute__((address_space(3))) float *LdsF;
typedef __attribute__((address_space(3))) int *LdsI;
__device__ int ReplayClosureWave(LdsF ncost, LdsI nlr, LdsI dstack, LdsI newq, LdsI lcode, LdsF lw, Arr<const int32_t> queue, int n_queue,
                                 float cutoff, int n_new, Blk &sh) {
  const int lane = KH_TIDX & 63;
  int cnt = 0;
#ifdef KH_X_STAMPS
#define RC(k, v) do { if (lane == 0) sh->phase[96 + 44 + (k)] += (v); } while (0)
#else
#define RC(k, v) do {} while (0)
#endif
  if (n_new == 0) return 0;   // (nothing to order: the pops only move costs, which the parallel closure has computed)
  for (int top = n_queue; top > 0; top -= 64) {   // uniform
    const int bsz = min(64, top);
    const int node = lane < bsz ? queue[top - 1 - lane] : -1;   // lane j holds the j-th pop from here
    unsigned long long pending = __ballot(lane < bsz);
    RC(0, 1);
    while (pending != 0ull) {
      RC(1, 1);
      bool act = false;
      if (((pending >> lane) & 1ull) != 0ull) {
        const float c = ncost[node];
        if (!(c > cutoff)) {
          const int r = nlr[node];
          for (int l = r & 0xffff; l < (r >> 16) && !act; l++) {
            const int code = lcode[l];
            if (code < 0) continue;
            const float tot = c + lw[l];
            if (!(tot < cutoff)) continue;
            if ((code & 0x40000000) != 0) {
              act = newq[code & 0x3fffffff] < 0;
            } else {
              const float cv = ncost[code & 0xfff];
              act = cv == INFINITY || tot < cv;
            }
          }
        }
      }
      const unsigned long long am = __ballot(act) & pending;
      if (am == 0ull) break;
      const int L = Uni(__ffsll(static_cast<long long>(am)) - 1);
      pending &= ~((2ull << L) - 1ull);
      const int first = __builtin_amdgcn_readlane(node, L);
      RC(2, 1);
      if (lane == 0) {   // the pop of `first` and everything it pushes, depth first
        int sp = 0;
        dstack[sp++] = first;
        while (sp > 0) {
          RC(3, 1);
          const int uu = dstack[--sp];
          const float c = ncost[uu];
          if (c > cutoff) continue;   // :779
          const int r = nlr[uu];
          for (int l = r & 0xffff; l < (r >> 16); l++) {
            const int code = lcode[l];
            if (code < 0) continue;
            const float tot = c + lw[l];
            if (!(tot < cutoff)) continue;   // :794
            if ((code & 0x40000000) != 0) {
              const int k = code & 0x3fffffff;
              if (newq[k] < 0) newq[k] = cnt++;
            } else {
              const int v = code & 0xfff;
              const float cv = ncost[v];
              bool push = false;
              if (cv == INFINITY) {
                newq[(code >> 12) - 1] = cnt++;
                push = true;
              } else if (tot < cv) {
                push = true;
              }
              if (push) {
                ncost[v] = tot;
                if (sp >= kRpNodes) { cnt = -1; sp = 0; break; }
                dstack[sp++] = v;
              }
            }
          }
        }
      }
      cnt = __builtin_amdgcn_readfirstlane(cnt);
      if (cnt < 0) return -1;
      if (cnt >= n_new) { RC(4, 1); RC(5, top); return cnt; }   // every token of the closure has its place: the rest of the queue only moves costs
    }
  }
  return cnt;
#undef RC
}

// Diagnostic: cycles since the previous SubStamp / Stamp of this workgroup into phase[ph] WITHOUT moving the phase timer
// (sub-phases of a phase that Stamp charges as a whole).
__device__ __forceinline__ void SubStamp(const Utt &u, Blk &sh, int ph) {
  if (u.phase_cycles != nullptr && KH_TIDX == 0) {
    const long long now = static_cast<long long>(__builtin_amdgcn_s_memtime());
    sh->phase[ph] += now - sh->t_sub;
    sh->t_sub = now;
  }
}

// Fine stamps of the reference-order phases (-DKH_X_STAMPS builds only): cycles since the previous XS / SubStamp into phase[64 + k].
#if defined(KH_X_STAMPS) && KH_X_STAMPS == 2
// (every wave first waits for its own outstanding memory operations and for the others: a step is charged the stores it issued)
#define XS(k) do { KhSync(); SubStamp(u, sh, 96 + (k)); } while (0)
#elif defined(KH_X_STAMPS)
#define XS(k) SubStamp(u, sh, 96 + (k))
#elif defined(KH_SERVE_MARKERS)
#define XS(k) KM(32 + (k), 0)
#else
#define XS(k) do {} while (0)
#endif

// List positions (x_pos) of the frame under construction, tokens [nb, fe): the emitting pass made [nb, ne_emit) with
// their insertion keys in x_q and, for those with epsilon arcs, their costs in x_c0e; the closure has converged and the frame's epsilon links are
// the block [lb, le).  Leaves x_bmin all ones.  Returns false on a capacity overflow (sh->status).
__device__ bool OrderFrontierSort(const Utt &u, const Params &p, int nb, int fe, int lb, int le, float cutoff, Blk &sh) {
  const int ne_emit = Uni(sh->x_ne_emit), n = fe - nb, eps_emit = Uni(sh->x_eps_emit), eps_n = Uni(sh->eps_n);
  const uint32_t H = Uni(sh->x_hsize), qbase = Uni(sh->x_qbase);
  const int nl = le - lb;
  if (u.phase_cycles != nullptr && KH_TIDX == 0) {
    sh->t_sub = static_cast<long long>(__builtin_amdgcn_s_memtime());
    sh->phase[47] += 1; sh->phase[52] += eps_emit; sh->phase[53] += eps_n; sh->phase[54] += nl; sh->phase[55] += fe - ne_emit; sh->phase[46] += n;
  }
  // buckets; first occupation among the emitting pass's tokens; closure replay tables
  {
    // (kOU tokens of a lane in flight together: the plain loop paid two dependent round trips - state, then the caller's
    // id - per trip, five trips per frame)
    constexpr int kOU = 4;
    for (int i0 = nb + KH_TIDX; i0 < fe; i0 += NT * kOU) {
      int32_t st[kOU], sid[kOU];
      uint32_t qv[kOU];
#pragma unroll
      for (int k = 0; k < kOU; k++) {
        const int i = min(i0 + k * NT, fe - 1);
        st[k] = u.tok_state[i];
        qv[k] = UX(x_q)[i - nb];
      }
#pragma unroll
      for (int k = 0; k < kOU; k++) sid[k] = -1 - p.unit_ilabel[st[k]];   // the caller's state id (what the reference hashes)
#pragma unroll
      for (int k = 0; k < kOU; k++) {
        const int i = i0 + k * NT;
        if (i >= fe) continue;
        const int32_t bk = static_cast<int32_t>(static_cast<uint32_t>(sid[k]) % H);
        UX(x_bkt)[i - nb] = bk;
        UX(x_epsidx)[i - nb] = -1;
        if (i < ne_emit) __hip_atomic_fetch_min(&UX(x_bmin)[bk], qv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else UX(x_q)[i - nb] = 0xFFFFFFFFu;
      }
    }
  }
  KhSync();
  for (int j = KH_TIDX; j < eps_n; j += NT) {
    const int tok = u.tmp_epslist[j];
    UX(x_epsidx)[tok - nb] = j;
    UX(x_nl0)[j] = 0;
    UX(x_nl1)[j] = 0;
    UX(x_ncost)[j] = j < eps_emit ? Dec(UX(x_c0e)[j]) : INFINITY;
    if (j < eps_emit) {   // (the first eps_emit entries are the emitting pass's: the queue's initial content, :766-767)
      UX(x_key0)[j] = (static_cast<unsigned long long>(__hip_atomic_load(&UX(x_bmin)[UX(x_bkt)[tok - nb]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) << 32) |
                    UX(x_q)[tok - nb];
      UX(x_val0)[j] = j;
    }
  }
  KhSync();
  for (int l = lb + KH_TIDX; l < le; l += NT) {
    const int src = u.link_src[l], dst = u.link_dst[l];
    const int j = UX(x_epsidx)[src - nb];
    if (l == lb || u.link_src[l - 1] != src) UX(x_nl0)[j] = l - lb;
    if (l + 1 == le || u.link_src[l + 1] != src) UX(x_nl1)[j] = l - lb + 1;
    int code = -1;
    if (dst >= 0) {
      const int e = UX(x_epsidx)[dst - nb];
      if (e >= 0) code = e;
      else if (dst >= ne_emit) code = 0x40000000 | (dst - nb);
    }
    UX(x_ord)[l - lb] = code;
    UX(x_lw)[l - lb] = __int_as_float(p.n_arcs[-1 - u.link_arc[l]].z);
  }
  const int bits = 32 - __clz(static_cast<int>(qbase + static_cast<uint32_t>(fe - ne_emit)) | 1);
  KhSync();
  SubStamp(u, sh, 48);
  // ---- the queue's initial content in list order -> x_stack[0, eps_emit)
  if (eps_emit <= 1) {
    if (KH_TIDX == 0 && eps_emit == 1) UX(x_stack)[0] = 0;
  } else {   // (ranking a few hundred keys by counting over an LDS copy measured 4 x slower than these four passes)
    const int sb = BlockRadixSort(u, eps_emit, bits, sh);
    const Arr<const int32_t> sorted = sb ? UX(x_val1) : UX(x_val0);
    for (int j = KH_TIDX; j < eps_emit; j += NT) UX(x_stack)[j] = sorted[j];
  }
  KhSync();
  SubStamp(u, sh, 49);
  // ---- closure replay: by one wave from LDS when the frame's epsilon structure fits (it nearly always does), else by
  // one lane from memory
  const int n_new = fe - ne_emit;
  bool in_lds = eps_n <= kRpNodes && nl <= kRpLinks && n_new <= kRpNodes;
  LdsF l_ncost = (LdsF)LdsKeys(sh);
  LdsI l_nlr = (LdsI)(LdsKeys(sh) + kRpNodes);
  LdsI l_dstack = (LdsI)(LdsKeys(sh) + 2 * kRpNodes);
  LdsI l_newq = (LdsI)(LdsKeys(sh) + 3 * kRpNodes);
  LdsI l_code = (LdsI)LdsVals(sh);
  LdsF l_lw = (LdsF)(LdsVals(sh) + kRpLinks);
  if (in_lds) {
    for (int j = KH_TIDX; j < eps_n; j += NT) {
      l_ncost[j] = UX(x_ncost)[j];
      l_nlr[j] = UX(x_nl0)[j] | (UX(x_nl1)[j] << 16);
    }
    for (int k = KH_TIDX; k < n_new; k += NT) l_newq[k] = -1;
    for (int l = KH_TIDX; l < nl; l += NT) {
      int code = UX(x_ord)[l];
      if (code >= 0) {
        if ((code & 0x40000000) != 0) {
          code = 0x40000000 | ((code & 0x3fffffff) - (ne_emit - nb));
        } else {
          const int tok = u.tmp_epslist[code];
          code |= (tok >= ne_emit ? tok - ne_emit + 1 : 0) << 12;
        }
      }
      l_code[l] = code;
      l_lw[l] = UX(x_lw)[l];
    }
    KhSync();
    if (KH_TIDX < 64) {
      const int cnt = ReplayClosureWave(l_ncost, l_nlr, l_dstack, l_newq, l_code, l_lw, UX(x_stack), eps_emit, cutoff, n_new, sh);
      if (KH_TIDX == 0) sh->x_n_new = cnt;
    }
    KhSync();
    in_lds = Uni(sh->x_n_new) >= 0;   // (-1: the depth-first stack outgrew LDS - from memory then)
    if (in_lds) {
      bool bad = false;
      for (int k = KH_TIDX; k < n_new; k += NT) {
        const int q = l_newq[k];
        bad |= q < 0;
        UX(x_q)[ne_emit - nb + k] = qbase + static_cast<uint32_t>(q);
      }
      // every token the closure created was inserted by the replay (the parallel fixed point and the queue reach the same set)
      if (bad || Uni(sh->x_n_new) != n_new) sh->status = 8;
    }
  }
  if (!in_lds) {
    KhSync();
    if (KH_TIDX == 0) {
      const int cnt = ReplayClosure((GP(float))UX(x_ncost).p, (GP(int32_t))UX(x_nl0).p, (GP(int32_t))UX(x_nl1).p, (GP(int32_t))UX(x_ord).p,
                                    (GP(float))UX(x_lw).p, (GP(int32_t))UX(x_stack).p, 0, u, sh, eps_emit, cutoff, nb, qbase, n_new);
      if (cnt != n_new) sh->status = cnt < 0 ? 3 : 8;
    }
  }
  KhSync();
  SubStamp(u, sh, 50);
  if (Uni(sh->status) != 0) return false;
  // ---- the closure's tokens enter their buckets; key = (first occupation of the bucket, insertion key); sort
  for (int i = ne_emit + KH_TIDX; i < fe; i += NT)
    __hip_atomic_fetch_min(&UX(x_bmin)[UX(x_bkt)[i - nb]], UX(x_q)[i - nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  KhSync();
  {
    constexpr int kOU = 4;
    for (int i0 = KH_TIDX; i0 < n; i0 += NT * kOU) {
      int32_t bk[kOU];
      uint32_t qv[kOU], bm[kOU];
#pragma unroll
      for (int k = 0; k < kOU; k++) {
        const int i = min(i0 + k * NT, n - 1);
        bk[k] = UX(x_bkt)[i];
        qv[k] = UX(x_q)[i];
      }
#pragma unroll
      for (int k = 0; k < kOU; k++) bm[k] = __hip_atomic_load(&UX(x_bmin)[bk[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int k = 0; k < kOU; k++) {
        const int i = i0 + k * NT;
        if (i >= n) continue;
        UX(x_key0)[i] = (static_cast<unsigned long long>(bm[k]) << 32) | qv[k];
        UX(x_val0)[i] = i;
      }
    }
  }
  KhSync();
  {
    constexpr int kOU = 4;
    for (int i0 = KH_TIDX; i0 < n; i0 += NT * kOU) {   // the table's invariant between frames
      int32_t bk[kOU];
#pragma unroll
      for (int k = 0; k < kOU; k++) bk[k] = UX(x_bkt)[min(i0 + k * NT, n - 1)];
#pragma unroll
      for (int k = 0; k < kOU; k++)
        if (i0 + k * NT < n) UX(x_bmin)[bk[k]] = 0xFFFFFFFFu;
    }
  }
  const int fb2 = n > 1 ? BlockRadixSort(u, n, bits, sh) : 0;
  KhSync();
  const Arr<const int32_t> order = fb2 ? UX(x_val1) : UX(x_val0);
  {
    constexpr int kOU = 8;
    for (int r0 = KH_TIDX; r0 < n; r0 += NT * kOU) {
      int32_t o[kOU];
#pragma unroll
      for (int k = 0; k < kOU; k++) o[k] = order[min(r0 + k * NT, n - 1)];
#pragma unroll
      for (int k = 0; k < kOU; k++)
        if (r0 + k * NT < n) UX(x_pos)[o[k]] = r0 + k * NT;
    }
  }
  KhSync();
  SubStamp(u, sh, 51);
  return true;
}

// ---------------------------------------------------------------- list order without a sort (round 5)
// The same list positions from DENSE RANKS instead of sorted 64-bit keys.  What a HashList position is made of:
//   r(t)   the rank of token t in the order of HashList::Insert calls: the emitting pass's tokens by the ordinal of their
//          first accepted arc (distinct integers below the frame's arc count: a BITMAP over the ordinals + a popcount
//          scan ranks them - no comparison sort), then the closure's tokens in the order the replay finds;
//   h(t)   the rank of the first token of t's bucket (state % hash_size), inb(t) t's rank inside its bucket;
//   pos(t) = #{t': h(t') < h(t)} + inb(t): buckets in the order of their first occupation, a bucket's tokens in
//          insertion order (hash-list-inl.h:118-147).
// Buckets: a second bitmap over the bucket ids gives every occupied bucket a dense index, a counting sort by that index
// lays the ranks of a bucket's tokens side by side, and a token reads its bucket's few members for h and inb.  The
// counts by h and ONE scan over the rank space turn (h, inb) into positions.  Everything lives in the two 32 KB LDS
// areas of the token table (idle between the closure and the next frame); the only global traffic is the per-token
// arrays, read and written coalesced.  The closure's initial queue (the emitting pass's tokens with epsilon arcs in
// list order) is ranked the same way - a bitmap over the positions among the emitting pass's tokens - and the few
// tokens the closure creates find their bucket mates among themselves by direct comparison.
// Frames beyond the LDS capacity (more than kFastN tokens, more than 2^18 buckets, more than kFastNew closure tokens)
// take OrderFrontierSort.
constexpr int kFastN = 65535, kFastNew = 1024, kFastMulti = 2 * kLdsSlots;

// The lane's index in the workgroup as a value the optimizer cannot see through.  The persistent kernel is one loop over
// frames around everything; with plain KH_TIDX the loop-invariant code motion hoists every `tid * 8 + j`, `w < W`
// and LDS address out of that loop, the 64-register budget cannot hold them, and each USE becomes a scratch_load with its
// own s_waitcnt vmcnt(0) - the round-5 listing of the 8-word LDS scans below had sixteen of those in a row (24 k cycles
// for a scan whose arithmetic takes a few hundred).  An opaque copy makes the address arithmetic a per-call VALU
// instruction again.
__device__ __forceinline__ int OpaqueTid() {
  int t = static_cast<int>(threadIdx.x);
  asm volatile("" : "+v"(t));
  return t;
}
// `const TidRef tid;` - a lane index that is re-made at every use (one v_mov) instead of living in a register from the top of
// a routine that spans a dozen barriers: as a plain `const int tid = OpaqueTid()` it was the most reloaded spill slot of the
// reference-order kernel after round 6's first pass (22 of its 90 static scratch loads).
struct TidRef {
  __device__ __forceinline__ operator int() const { return OpaqueTid(); }
};

// exclusive prefix of the set-bit counts of bits[0, W) -> pre[0, W), W <= kLdsSlots; returns the number of set bits.
// One barrier inside; the caller syncs before it reads pre.
__device__ __forceinline__ int BitmapPrefix(LdsU32 bits, LdsU32 pre, int W, Blk &sh) {
  const TidRef tid;
  constexpr int kPer = kLdsSlots / NT;
  int c[kPer], mine = 0;
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    const int w = tid * kPer + j;
    const uint32_t v = bits[w];   // (unguarded: inside the area whatever W is - the eight reads go out together)
    c[j] = w < W ? __popc(v) : 0;
    mine += c[j];
  }
  int total;
  int run = BlockExScan<true>(mine, &total, sh);
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    const int w = tid * kPer + j;
    if (w < W) pre[w] = static_cast<uint32_t>(run);
    run += c[j];
  }
  return total;
}
// a[0, W) -> its exclusive prefix sums, in place (a lane owns kLdsSlots / NT consecutive words); returns the total
__device__ __forceinline__ int LdsExScanInPlace(LdsU32 a, int W, Blk &sh) {
  const TidRef tid;
  constexpr int kPer = kLdsSlots / NT;
  int c[kPer], mine = 0;
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    const int w = tid * kPer + j;
    const uint32_t v = a[w];   // (unguarded: inside the area whatever W is - the eight reads go out together)
    c[j] = w < W ? static_cast<int>(v) : 0;
    mine += c[j];
  }
  int total;
  int run = BlockExScan<true>(mine, &total, sh);
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    const int w = tid * kPer + j;
    if (w < W) a[w] = static_cast<uint32_t>(run);
    run += c[j];
  }
  return total;
}
__device__ __forceinline__ uint32_t LdsLoadU(LdsU32 p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int BitRank(LdsU32 bits, LdsU32 pre, uint32_t q) {   // set bits below bit q
  return static_cast<int>(pre[q >> 5]) + __popc(bits[q >> 5] & ((1u << (q & 31u)) - 1u));
}

// The closure replay's tables in global memory (frames whose epsilon structure does not fit LDS, or whose depth-first
// stack outgrew it) and the replay by one lane.  x_epsidx must hold -1 for every token of the frame.
__device__ void ReplayFromMemory(const Utt &u, const Params &p, int nb, int fe, int lb, int le, float cutoff, uint32_t qbase, Blk &sh) {
  const int ne_emit = Uni(sh->x_ne_emit), eps_emit = Uni(sh->x_eps_emit), eps_n = Uni(sh->eps_n), n_new = fe - ne_emit;
  for (int j = KH_TIDX; j < eps_n; j += NT) {
    const int tok = u.tmp_epslist[j];
    UX(x_epsidx)[tok - nb] = j;
    UX(x_nl0)[j] = 0;
    UX(x_nl1)[j] = 0;
    UX(x_ncost)[j] = j < eps_emit ? Dec(UX(x_c0e)[j]) : INFINITY;
  }
  for (int i = ne_emit + KH_TIDX; i < fe; i += NT) UX(x_q)[i - nb] = 0xFFFFFFFFu;
  KhSync();
  for (int l = lb + KH_TIDX; l < le; l += NT) {
    const int src = u.link_src[l], dst = u.link_dst[l];
    const int j = UX(x_epsidx)[src - nb];
    if (l == lb || u.link_src[l - 1] != src) UX(x_nl0)[j] = l - lb;
    if (l + 1 == le || u.link_src[l + 1] != src) UX(x_nl1)[j] = l - lb + 1;
    int code = -1;
    if (dst >= 0) {
      const int e = UX(x_epsidx)[dst - nb];
      if (e >= 0) code = e;
      else if (dst >= ne_emit) code = 0x40000000 | (dst - nb);
    }
    UX(x_ord)[l - lb] = code;
    UX(x_lw)[l - lb] = __int_as_float(p.n_arcs[-1 - u.link_arc[l]].z);
  }
  KhSync();
  if (KH_TIDX == 0) {
    const int cnt = ReplayClosure((GP(float))UX(x_ncost).p, (GP(int32_t))UX(x_nl0).p, (GP(int32_t))UX(x_nl1).p, (GP(int32_t))UX(x_ord).p,
                                  (GP(float))UX(x_lw).p, (GP(int32_t))UX(x_stack).p, 0, u, sh, eps_emit, cutoff, nb, qbase, n_new);
    if (cnt != n_new) sh->status = cnt < 0 ? 3 : 8;
  }
  KhSync();
}

// pos(t) = #{t': h(t') < h(t)} + inb(t) for the tokens [0, cnt) of the frame under construction -> x_pos.  One count per
// first-of-bucket rank and one scan over the rank space, 8192 ranks at a time.
__device__ void PositionsFromHeads(const Utt &u, int cnt, Blk &sh, int xs0) {
  const TidRef tid;
  const LdsU32 K = (LdsU32)LdsKeys(sh);
  constexpr int kU = 8;   // (a lane's loads of eight trips in flight together: a loop that waits per trip pays a memory round trip per trip)
  int pbase = 0;
  for (int h0 = 0; h0 < cnt; h0 += kLdsSlots) {
    const uint32_t W = static_cast<uint32_t>(min(kLdsSlots, cnt - h0));
    for (uint32_t w = tid; w < W; w += NT) K[w] = 0u;
    LdsSync();
    XS(xs0 + 0);
    for (int i0 = tid; i0 < cnt; i0 += NT * kU) {
      int hh[kU];
#pragma unroll
      for (int k = 0; k < kU; k++) hh[k] = UX(x_h)[min(i0 + k * NT, cnt - 1)];
#pragma unroll
      for (int k = 0; k < kU; k++) {
        const uint32_t at = static_cast<uint32_t>(hh[k] - h0);
        if (i0 + k * NT < cnt && at < W) (void)__hip_atomic_fetch_add(&K[at], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    LdsSync();
    XS(xs0 + 1);
    const int total = LdsExScanInPlace(K, static_cast<int>(W), sh);
    LdsSync();
    XS(xs0 + 2);
    for (int i0 = tid; i0 < cnt; i0 += NT * kU) {
      int hh[kU], ib[kU];
#pragma unroll
      for (int k = 0; k < kU; k++) {
        const int i = min(i0 + k * NT, cnt - 1);
        hh[k] = UX(x_h)[i];
        ib[k] = UX(x_inb)[i];
      }
#pragma unroll
      for (int k = 0; k < kU; k++) {
        const uint32_t at = static_cast<uint32_t>(hh[k] - h0);
        if (i0 + k * NT < cnt && at < W) UX(x_pos)[i0 + k * NT] = pbase + static_cast<int>(K[at]) + ib[k];
      }
    }
    pbase += total;
    LdsSync();
    XS(xs0 + 3);
  }
}

// Returns 1 = done, 0 = failed (sh->status), 2 = not applicable after all (more than kFastMulti tokens share their
// bucket with another one: nothing the sort needs has been touched - it takes the frame).
__device__ int OrderFrontierFast(const Utt &u, const Params &p, int nb, int fe, int lb, int le, float cutoff, Blk &sh) {
  const TidRef tid;
  const int ne_emit = Uni(sh->x_ne_emit), n = fe - nb, n_emit = ne_emit - nb, eps_emit = Uni(sh->x_eps_emit), eps_n = Uni(sh->eps_n);
  const uint32_t H = Uni(sh->x_hsize), qbase = Uni(sh->x_qbase);
  const int nl = le - lb, n_new = fe - ne_emit;
  const LdsU32 K = (LdsU32)LdsKeys(sh), V = (LdsU32)LdsVals(sh);
  const LdsU16 V16 = (LdsU16)LdsVals(sh);
  const int Hw = static_cast<int>((H + 31u) >> 5), qw = static_cast<int>((qbase + 31u) >> 5);
  const Arr<int32_t> xr = UX(x_val0);   // insertion ranks
  constexpr int kU = 8;   // (a lane's loads of eight trips in flight together)
  if (u.phase_cycles != nullptr && tid == 0) {
    sh->t_sub = static_cast<long long>(__builtin_amdgcn_s_memtime());
    sh->phase[47] += 1; sh->phase[52] += eps_emit; sh->phase[53] += eps_n; sh->phase[54] += nl; sh->phase[55] += n_new; sh->phase[46] += n;
  }
  // ---- 1. three bitmaps: the ordinals of the emitting pass's tokens (their ranks), the occupied buckets, the buckets
  // that hold more than one token (a token that finds its bucket's bit set, and a token of the closure whose bucket is
  // occupied, set the second bit).  A token alone in its bucket is done here: h = its own rank, inb = 0.
  const bool comb = qw + 2 * Hw <= kLdsSlots;
  const int o1 = comb ? qw : 0, o2 = o1 + Hw, W1 = o2 + Hw;
  if (!comb) {   // the ordinals alone, 2^18 at a time
    int rbase = 0;
    for (uint32_t c0 = 0; c0 < qbase; c0 += static_cast<uint32_t>(kLdsSlots) * 32u) {
      const int W = min(kLdsSlots, static_cast<int>((qbase - c0 + 31u) >> 5));
      for (int w = tid; w < W; w += NT) K[w] = 0u;
      LdsSync();
      XS(0);
      for (int i = tid; i < n_emit; i += NT) {
        const uint32_t q = UX(x_q)[i] - c0;
        if (q < static_cast<uint32_t>(kLdsSlots) * 32u) (void)__hip_atomic_fetch_or(&K[q >> 5], 1u << (q & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      LdsSync();
      XS(1);
      const int total = BitmapPrefix(K, V, W, sh);
      LdsSync();
      XS(2);
      for (int i = tid; i < n_emit; i += NT) {
        const uint32_t q = UX(x_q)[i] - c0;
        if (q < static_cast<uint32_t>(kLdsSlots) * 32u) xr[i] = rbase + BitRank(K, V, q);
      }
      rbase += total;
      LdsSync();
      XS(3);
    }
  }
  for (int w = tid; w < W1; w += NT) K[w] = 0u;
  LdsSync();
  XS(4);
  for (int i0 = tid; i0 < n; i0 += NT * kU) {
    int32_t sid[kU];
    uint32_t qv[kU];
    // the caller's state id (what the reference hashes): the emitting pass's tokens got it from their candidates (x_bkt), the
    // few tokens of the closure look it up (unit id -> label table)
#pragma unroll
    for (int k = 0; k < kU; k++) {
      const int i = min(i0 + k * NT, n - 1);
      sid[k] = i < n_emit ? UX(x_bkt)[i] : -1 - p.unit_ilabel[u.tok_state[nb + i]];
      qv[k] = UX(x_q)[i];
    }
#pragma unroll
    for (int k = 0; k < kU; k++) {
      const int i = i0 + k * NT;
      if (i >= n) continue;
      const uint32_t bk = static_cast<uint32_t>(sid[k]) % H;
      UX(x_bkt)[i] = static_cast<int32_t>(bk);
      UX(x_epsidx)[i] = -1;
      if (i < n_emit) {
        const uint32_t m = 1u << (bk & 31u);
        const uint32_t old = __hip_atomic_fetch_or(&K[o1 + (bk >> 5)], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((old & m) != 0u) (void)__hip_atomic_fetch_or(&K[o2 + (bk >> 5)], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (comb) (void)__hip_atomic_fetch_or(&K[qv[k] >> 5], 1u << (qv[k] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  LdsSync();
  XS(5);
  // (the closure's tokens, by the lane that stored their bucket id above: i = lane + a multiple of NT)
  for (int i = n_emit + static_cast<int>((tid + NT - n_emit % NT) % NT); i < n; i += NT) {
    const uint32_t bk = static_cast<uint32_t>(UX(x_bkt)[i]), m = 1u << (bk & 31u);
    if ((K[o1 + (bk >> 5)] & m) != 0u) (void)__hip_atomic_fetch_or(&K[o2 + (bk >> 5)], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  LdsSync();
  XS(6);
  const int n_bits = BitmapPrefix(K, V, W1, sh);
  LdsSync();
  XS(7);
  const int pre2 = static_cast<int>(V[o2]);   // set bits in front of the third bitmap
  const int n_b2 = n_bits - pre2;             // buckets with more than one token
  if (n_b2 > kLdsSlots) return 2;
  for (int i0 = tid; i0 < n; i0 += NT * kU) {
    uint32_t bks[kU], qv[kU];
#pragma unroll
    for (int k = 0; k < kU; k++) {
      const int i = min(i0 + k * NT, n - 1);
      bks[k] = static_cast<uint32_t>(UX(x_bkt)[i]);   // (this lane's own store)
      qv[k] = (comb || i >= n_emit) ? UX(x_q)[i] : static_cast<uint32_t>(xr[i]);
    }
#pragma unroll
    for (int k = 0; k < kU; k++) {
      const int i = i0 + k * NT;
      if (i >= n) continue;
      const uint32_t bk = bks[k], m = 1u << (bk & 31u);
      const int w2 = o2 + static_cast<int>(bk >> 5);
      const uint32_t bits2 = K[w2];
      const int r = i < n_emit ? (comb ? BitRank(K, V, qv[k]) : static_cast<int>(qv[k])) : -1;
      if (comb && i < n_emit) xr[i] = r;
      if ((bits2 & m) != 0u) {
        UX(x_bkt)[i] = static_cast<int>(V[w2]) - pre2 + __popc(bits2 & (m - 1u));   // dense index among the shared buckets
      } else {
        // alone: the token's own rank heads its bucket.  A token of the closure whose bucket holds none of the
        // emitting pass's tokens keeps its bucket id, as -2 - id (its mates are among the closure's tokens).
        UX(x_bkt)[i] = i < n_emit ? -1 : -2 - static_cast<int32_t>(bk);
        UX(x_h)[i] = r;
        UX(x_inb)[i] = 0;
      }
    }
  }
  LdsSync();
  XS(8);
  // ---- 2. the tokens of the shared buckets: counting sort by bucket index, a bucket's ranks side by side (16 bits each)
  if (n_b2 > 0) {
    for (int w = tid; w < n_b2; w += NT) K[w] = 0u;
    LdsSync();
    XS(9);
    for (int i0 = tid; i0 < n_emit; i0 += NT * kU) {
      int ds[kU];
#pragma unroll
      for (int k = 0; k < kU; k++) ds[k] = UX(x_bkt)[min(i0 + k * NT, n_emit - 1)];
#pragma unroll
      for (int k = 0; k < kU; k++)
        if (i0 + k * NT < n_emit && ds[k] >= 0) (void)__hip_atomic_fetch_add(&K[ds[k]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    LdsSync();
    XS(10);
    const int n_multi = LdsExScanInPlace(K, n_b2, sh);
    LdsSync();
    XS(11);
    if (n_multi > kFastMulti) return 2;
    for (int i0 = tid; i0 < n_emit; i0 += NT * kU) {
      int ds[kU], rs[kU];
#pragma unroll
      for (int k = 0; k < kU; k++) {
        const int i = min(i0 + k * NT, n_emit - 1);
        ds[k] = UX(x_bkt)[i];
        rs[k] = xr[i];
      }
#pragma unroll
      for (int k = 0; k < kU; k++)
        if (i0 + k * NT < n_emit && ds[k] >= 0)
          V16[__hip_atomic_fetch_add(&K[ds[k]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = static_cast<uint16_t>(rs[k]);
    }
    LdsSync();   // K[d] = end of bucket d's run = start of bucket d + 1's
    XS(12);
    for (int i0 = tid; i0 < n; i0 += NT * kU) {
      int ds[kU], rs[kU];
#pragma unroll
      for (int k = 0; k < kU; k++) {
        const int i = min(i0 + k * NT, n - 1);
        ds[k] = UX(x_bkt)[i];
        rs[k] = xr[i];
      }
#pragma unroll
      for (int k = 0; k < kU; k++) {
        const int i = i0 + k * NT, d = ds[k];
        if (i >= n || d < 0) continue;
        const int s0 = d > 0 ? static_cast<int>(K[d - 1]) : 0, s1 = static_cast<int>(K[d]);
        const uint32_t r = i < n_emit ? static_cast<uint32_t>(rs[k]) : 0xFFFFFFFFu;   // (a token of the closure comes after all of them)
        uint32_t hm = 0xFFFFFFFFu;
        int inb = 0;
        for (int m = s0; m < s1; m++) {
          const uint32_t rr = V16[m];
          hm = rr < hm ? rr : hm;
          inb += rr < r ? 1 : 0;
        }
        UX(x_h)[i] = static_cast<int>(hm);
        UX(x_inb)[i] = inb;
      }
    }
  }
  KhSync();   // (x_h / x_inb / x_epsidx = -1 have reached memory: other lanes read them from here on)
  XS(13);
  if (u.phase_cycles != nullptr && tid == 0) sh->phase[56] += 1;
  SubStamp(u, sh, 48);
  if (n_new > 0) {
    // ---- 3. the closure's initial queue: the emitting pass's tokens with epsilon arcs, in the order of the list as it is
    // BEFORE the closure inserts anything (:766-767) -> x_stack[0, eps_emit) (entries of tmp_epslist).  A few hundred
    // tokens: their (first-of-bucket rank, rank inside the bucket) pairs go to LDS and every token counts the smaller ones
    // (all lanes read the same words: broadcasts).  l_dstack[entry] = its pop (the queue is popped from the back).
    LdsI l_dstack = (LdsI)(LdsKeys(sh) + 2 * kRpNodes);
    if (eps_emit <= kRpNodes) {
      for (int j = tid; j < eps_emit; j += NT) {
        const int i = u.tmp_epslist[j] - nb;
        V[j] = (static_cast<uint32_t>(UX(x_h)[i]) << 16) | static_cast<uint32_t>(UX(x_inb)[i]);
      }
      LdsSync();
      XS(14);
      for (int j = tid; j < eps_emit; j += NT) {
        const uint32_t key = V[j];
        int rank = 0;
        int j2 = 0;
        for (; j2 + 4 <= eps_emit; j2 += 4) {
          const uint32_t a0 = V[j2], a1 = V[j2 + 1], a2 = V[j2 + 2], a3 = V[j2 + 3];
          rank += (a0 < key ? 1 : 0) + (a1 < key ? 1 : 0) + (a2 < key ? 1 : 0) + (a3 < key ? 1 : 0);
        }
        for (; j2 < eps_emit; j2++) rank += V[j2] < key ? 1 : 0;
        UX(x_stack)[rank] = j;
        l_dstack[j] = eps_emit - 1 - rank;
      }
    } else if (eps_emit > 1) {
      PositionsFromHeads(u, n_emit, sh, 20);   // positions among the emitting pass's tokens
      KhSync();
      const int pw = (n_emit + 31) >> 5;   // (<= 2048 words)
      for (int w = tid; w < pw; w += NT) V[w] = 0u;
      LdsSync();
      for (int j = tid; j < eps_emit; j += NT) {
        const uint32_t pe = static_cast<uint32_t>(UX(x_pos)[u.tmp_epslist[j] - nb]);
        (void)__hip_atomic_fetch_or(&V[pe >> 5], 1u << (pe & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      LdsSync();
      (void)BitmapPrefix(V, V + kLdsSlots / 2, pw, sh);
      LdsSync();
      for (int j = tid; j < eps_emit; j += NT) {
        const uint32_t pe = static_cast<uint32_t>(UX(x_pos)[u.tmp_epslist[j] - nb]);
        UX(x_stack)[BitRank(V, V + kLdsSlots / 2, pe)] = j;
      }
    }
    KhSync();
    XS(18);
    SubStamp(u, sh, 49);
    // ---- 4. closure replay: its tables straight into LDS when the frame's epsilon structure fits (it nearly always does)
    bool in_lds = eps_n <= kRpNodes && nl <= kRpLinks && n_new <= kRpNodes;
    LdsF l_ncost = (LdsF)LdsKeys(sh);
    LdsI l_nlr = (LdsI)(LdsKeys(sh) + kRpNodes);
    LdsI l_newq = (LdsI)(LdsKeys(sh) + 3 * kRpNodes);
    LdsI l_code = (LdsI)LdsVals(sh);
    LdsF l_lw = (LdsF)(LdsVals(sh) + kRpLinks);
    if (in_lds) {
      for (int j = tid; j < eps_n; j += NT) {
        const int tok = u.tmp_epslist[j];
        UX(x_epsidx)[tok - nb] = j;
        l_ncost[j] = j < eps_emit ? Dec(UX(x_c0e)[j]) : INFINITY;
        l_nlr[j] = 0;
      }
      for (int k = tid; k < n_new; k += NT) l_newq[k] = -1;
      KhSync();
      XS(19);
      constexpr int kLU = 4;
      for (int l0 = lb + tid; l0 < le; l0 += NT * kLU) {
        int srcs[kLU], dsts[kLU], arcs[kLU], prevs[kLU], nexts[kLU], js[kLU], es[kLU];
        float ws[kLU];
#pragma unroll
        for (int k = 0; k < kLU; k++) {
          const int l = min(l0 + k * NT, le - 1);
          srcs[k] = u.link_src[l];
          dsts[k] = u.link_dst[l];
          arcs[k] = u.link_arc[l];
          prevs[k] = l > lb ? u.link_src[l - 1] : -1;
          nexts[k] = l + 1 < le ? u.link_src[l + 1] : -1;
        }
#pragma unroll
        for (int k = 0; k < kLU; k++) {
          js[k] = UX(x_epsidx)[srcs[k] - nb];
          es[k] = dsts[k] >= 0 ? UX(x_epsidx)[dsts[k] - nb] : -1;
          ws[k] = __int_as_float(p.n_arcs[-1 - arcs[k]].z);
        }
#pragma unroll
        for (int k = 0; k < kLU; k++) {
          const int l = l0 + k * NT;
          if (l >= le) continue;
          int rng = 0;
          if (prevs[k] != srcs[k]) rng |= l - lb;
          if (nexts[k] != srcs[k]) rng |= (l - lb + 1) << 16;
          if (rng != 0) (void)__hip_atomic_fetch_or(&l_nlr[js[k]], rng, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          int code = -1;
          if (dsts[k] >= 0) {
            if (es[k] >= 0) code = es[k] | ((dsts[k] >= ne_emit ? dsts[k] - ne_emit + 1 : 0) << 12);
            else if (dsts[k] >= ne_emit) code = 0x40000000 | (dsts[k] - ne_emit);
          }
          l_code[l - lb] = code;
          l_lw[l - lb] = ws[k];
        }
      }
      LdsSync();
      XS(30);
      // ---- insertion order without the queue, when the frame's epsilon structure allows it (it does on an HCLG: the
      // tokens epsilon arcs lead to - language-model states - are all made by the closure, none by the emitting pass).
      // If no processed link leads to a token of the initial queue, the queue's tokens keep their costs while it runs,
      // and a token X of the closure is inserted at the FIRST pop p(u) of a queue token u from which a path leads to X
      // on which every total stays under the cutoff: an arrival that finds a token on the way already as cheap was
      // preceded by one that went further with totals at least as good.  So every live link of a queue token walks
      // down its chain (tokens of the closure with ONE live link each) and leaves (pop, depth, first link of the walk)
      // at every token it reaches, minimum wins; the insertion order is the order of those keys - siblings of one pop in
      // arc order, then ONE chain by depth.  Anything else (a link into the initial queue, a closure token with two
      // live links, two chains of one pop with tokens beyond their first, a walk of 30 steps) leaves the frame to the
      // interpreter of the LIFO queue below.  ~33 insertions per frame took it ~50 us (round-5 stamps), one lane
      // chasing LDS round trips; this takes a few barriers.
      bool by_walks = false;
#ifndef KH_X_NO_WALKS
      if (n_new < kRpNodes / 2 && eps_emit <= kRpNodes) {
        typedef __attribute__((address_space(3))) unsigned long long *LdsU64;
        // [n_new] over newq (written last); the area starts on a 4-byte boundary: the 64-bit atomics need 8
        const LdsU64 key64 = (LdsU64)(l_newq + ((reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void *)l_newq) >> 2) & 1u));
        for (int k = tid; k < n_new; k += NT) key64[k] = ~0ull;
        if (tid == 0) sh->flag = 0;
        LdsSync();
        for (int j = tid; j < eps_emit; j += NT) {
          const float c0 = l_ncost[j];
          if (c0 > cutoff) continue;   // :779
          const unsigned long long pop = static_cast<unsigned long long>(l_dstack[j]) << 40;
          const int r = l_nlr[j];
          for (int l = r & 0xffff; l < (r >> 16); l++) {
            int code = l_code[l];
            if (code < 0) continue;
            float tot = c0 + l_lw[l];
            if (!(tot < cutoff)) continue;   // :794
            for (unsigned long long depth = 1;; depth++) {
              const bool leaf = (code & 0x40000000) != 0;
              const int k = leaf ? (code & 0x3fffffff) : (code >> 12) - 1;
              if (k < 0 || depth > 30) { sh->flag = 1; break; }   // a token of the emitting pass with epsilon arcs: it sits in the queue
              (void)__hip_atomic_fetch_min(&key64[k], pop | (depth << 32) | static_cast<unsigned long long>(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (leaf) break;
              const int r2 = l_nlr[code & 0xfff];
              int l2 = -1;
              for (int ll = r2 & 0xffff; ll < (r2 >> 16); ll++)
                if (l_code[ll] >= 0) { if (l2 >= 0) sh->flag = 1; l2 = ll; }
              if (l2 < 0) break;
              tot = tot + l_lw[l2];
              if (!(tot < cutoff)) break;
              code = l_code[l2];
            }
          }
        }
        LdsSync();
        int my_rank = 0;
        bool bad = false;
        const int k = tid;   // (n_new <= 1024: a lane per token)
        if (k < n_new) {
          const unsigned long long mine = key64[k];
          bad = mine == ~0ull;
          const unsigned long long my_pop = mine >> 40, my_depth = (mine >> 32) & 0xffull, my_chain = mine & 0xffffffffull;
          for (int k2 = 0; k2 < n_new; k2++) {
            const unsigned long long o = key64[k2];
            my_rank += o < mine ? 1 : 0;
            if (k2 != k && (o >> 40) == my_pop) {
              const unsigned long long od = (o >> 32) & 0xffull;
              if (od == my_depth && (my_depth >= 2 || o == mine)) bad = true;                      // two tokens of one pop at one depth beyond the first
              if (my_depth >= 2 && od >= 2 && (o & 0xffffffffull) != my_chain) bad = true;       // two chains of one pop
            }
          }
        }
        if (bad) sh->flag = 1;
        LdsSync();
        by_walks = Uni(sh->flag) == 0;
        LdsSync();   // (every lane has read the keys and the flag)
        if (by_walks) {
          if (k < n_new) l_newq[k] = my_rank;
          if (tid == 0) sh->x_n_new = n_new;
        } else {
          for (int k3 = tid; k3 < n_new; k3 += NT) l_newq[k3] = -1;
        }
        if (u.phase_cycles != nullptr && tid == 0) sh->phase[by_walks ? 64 : 65] += 1;
        LdsSync();
      }
#endif
      if (!by_walks && tid < 64) {
        const int cnt = ReplayClosureWave(l_ncost, l_nlr, l_dstack, l_newq, l_code, l_lw, UX(x_stack), eps_emit, cutoff, n_new, sh);
        if (tid == 0) sh->x_n_new = cnt;
      }
      LdsSync();
      XS(31);
      in_lds = Uni(sh->x_n_new) >= 0;   // (-1: the depth-first stack outgrew LDS - from memory then)
      if (in_lds) {
        bool bad = false;
        for (int k = tid; k < n_new; k += NT) {
          const int q = l_newq[k];
          bad |= q < 0;
          xr[n_emit + k] = n_emit + q;
        }
        // every token the closure created was inserted by the replay (the parallel fixed point and the queue reach the same set)
        if (bad || Uni(sh->x_n_new) != n_new) sh->status = 8;
      }
    }
    if (!in_lds) {
      KhSync();
      XS(32);
      ReplayFromMemory(u, p, nb, fe, lb, le, cutoff, qbase, sh);
      for (int k = tid; k < n_new; k += NT) xr[n_emit + k] = n_emit + static_cast<int>(UX(x_q)[n_emit + k] - qbase);
    }
    LdsSync();
    XS(33);
    SubStamp(u, sh, 50);
    if (Uni(sh->status) != 0) return 0;
    // ---- 5. the closure's tokens find their bucket mates among themselves: the same x_bkt value = the same bucket
    // (>= 0: a bucket that holds tokens of the emitting pass - they come behind those; <= -2: a bucket of their own)
    for (int k = tid; k < n_new; k += NT) {
      V[k] = static_cast<uint32_t>(UX(x_bkt)[n_emit + k]);
      V[kFastNew + k] = static_cast<uint32_t>(xr[n_emit + k]);   // (this lane's own store)
    }
    LdsSync();
    XS(34);
    for (int k = tid; k < n_new; k += NT) {
      const uint32_t key = V[k], r = V[kFastNew + k];
      uint32_t minr = r;
      int before = 0;
      for (int k2 = 0; k2 < n_new; k2++) {
        if (V[k2] != key) continue;
        const uint32_t r2 = V[kFastNew + k2];
        before += r2 < r ? 1 : 0;
        minr = r2 < minr ? r2 : minr;
      }
      const int i = n_emit + k;
      if (static_cast<int32_t>(key) >= 0) {
        UX(x_inb)[i] += before;   // behind the bucket's tokens of the emitting pass (x_h: their first)
      } else {
        UX(x_h)[i] = static_cast<int>(minr);
        UX(x_inb)[i] = before;
      }
    }
    KhSync();
    XS(35);
  } else {
    SubStamp(u, sh, 49);
    SubStamp(u, sh, 50);
  }
  // ---- 6. positions
  PositionsFromHeads(u, n, sh, 40);
  KhSync();
  XS(36);
  SubStamp(u, sh, 51);
  return 1;
}

// The same construction with every per-token intermediate in LDS (round 5, second half).  The kernel in reference order
// moved 5.9 TB per launch against the canonical kernel's 2.6 TB and ran at the fabric's limit (rocprofv3 FETCH_SIZE /
// WRITE_SIZE); the list order's share was its per-token arrays - bucket, rank, first-of-bucket rank, rank inside the bucket -
// written to memory by one step and read back by the next, ~0.5 MB per frame (nothing survives in L2 between the phases of
// 64 workgroups per XCD).  For a frame of up to 8192 tokens ONE word per token stays in the dynamic LDS area (T), the
// bitmaps, the counting sort and the closure's tables share the static one; memory sees the inputs (state id, ordinal: one
// coalesced read each, the ordinals twice) and the positions.  Returns 1 = done, 2 = not applicable (a capacity below is
// exceeded, or the closure's order needs the queue's interpreter): nothing has been written - OrderFrontierFast takes the frame.
//   T[i]: step 2 bucket id -> step 4 rank | shared-bucket index << 13 | shared << 26 -> step 5 on: first-of-bucket rank (13
//   bits) | rank inside the bucket (8 bits) << 13 | (entry of tmp_epslist + 1) << 21.
constexpr int kT1Nodes = 1024, kT1Links = 1024, kT1New = 511;
__device__ int OrderFrontierLds(const Utt &u, const Params &p, int nb, int fe, int lb, int le, float cutoff, Blk &sh) {
  const TidRef tid;
  const int ne_emit = Uni(sh->x_ne_emit), n = fe - nb, n_emit = ne_emit - nb, eps_emit = Uni(sh->x_eps_emit), eps_n = Uni(sh->eps_n);
  const uint32_t H = Uni(sh->x_hsize), qbase = Uni(sh->x_qbase);
  const int nl = le - lb, n_new = fe - ne_emit;
  const int Hw = static_cast<int>((H + 31u) >> 5), qw = static_cast<int>((qbase + 31u) >> 5);
  // Bitmaps: the ordinals (qw words) need their prefix counts, the shared buckets (Hw) too, the occupied buckets (Hw) do not.
  // Side by side when everything fits the static area; else the ordinals first (the rank joins the bucket id in T: 17 + 13
  // bits), then the two bucket bitmaps - the table never shrinks (:219-225), so an utterance that once held a 40 k-token
  // frame keeps 2 x 2500 words of bucket bits for good.
  const bool comb = 2 * (qw + 2 * Hw) <= kLdsSlots;
  if (n > kLdsSlots || (!comb && (2 * qw > kLdsSlots || 3 * Hw > kLdsSlots || H > (1u << 17))) || eps_n > kT1Nodes || nl > kT1Links || n_new > kT1New)
    return 2;
  const LdsU32 K = (LdsU32)LdsKeys(sh), T = (LdsU32)LdsVals(sh);
  constexpr int kU = 8;
  if (u.phase_cycles != nullptr && tid == 0) sh->t_sub = static_cast<long long>(__builtin_amdgcn_s_memtime());
  // word offsets of the ordinal bits, the occupied-bucket bits, the shared-bucket bits; P / P2: prefix counts of the first / last
  const int o1 = comb ? qw : 0, o2 = o1 + Hw, W1 = o2 + Hw;
  const LdsU32 P = comb ? K + W1 : K + qw;          // prefix counts of the ordinal bits (comb: of all three bitmaps)
  const LdsU32 P2 = comb ? K + W1 + o2 : K + 2 * Hw;   // ... of the shared-bucket bits, by word of that bitmap
  if (tid == 0) sh->flag = 0;
  if (!comb) {
    // ---- 1a. the ordinals alone: ranks into T next to the bucket id
    for (int w = tid; w < qw; w += NT) K[w] = 0u;
    LdsSync();
    for (int i0 = tid; i0 < n; i0 += NT * kU) {
      int32_t sid[kU];
      uint32_t qv[kU];
#pragma unroll
      for (int k = 0; k < kU; k++) {
        const int i = min(i0 + k * NT, n - 1);
        sid[k] = i < n_emit ? UX(x_bkt)[i] : -1 - p.unit_ilabel[u.tok_state[nb + i]];
        qv[k] = UX(x_q)[i];
      }
#pragma unroll
      for (int k = 0; k < kU; k++) {
        const int i = i0 + k * NT;
        if (i >= n) continue;
        T[i] = static_cast<uint32_t>(sid[k]) % H;
        if (i < n_emit) (void)__hip_atomic_fetch_or(&K[qv[k] >> 5], 1u << (qv[k] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    LdsSync();
    (void)BitmapPrefix(K, P, qw, sh);
    LdsSync();
    for (int i = tid; i < n_emit; i += NT) T[i] |= static_cast<uint32_t>(BitRank(K, P, UX(x_q)[i])) << 17;
    LdsSync();
    // ---- 1b. the bucket bitmaps
    for (int w = tid; w < 2 * Hw; w += NT) K[w] = 0u;
    LdsSync();
    for (int i = tid; i < n_emit; i += NT) {
      const uint32_t bk = T[i] & 0x1ffffu, m = 1u << (bk & 31u);
      const uint32_t old = __hip_atomic_fetch_or(&K[o1 + (bk >> 5)], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if ((old & m) != 0u) (void)__hip_atomic_fetch_or(&K[o2 + (bk >> 5)], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  } else {
    // ---- 1-2. bitmaps (ordinals, occupied buckets, shared buckets); T = bucket id
    for (int w = tid; w < W1; w += NT) K[w] = 0u;
    LdsSync();
    for (int i0 = tid; i0 < n; i0 += NT * kU) {
      int32_t sid[kU];
      uint32_t qv[kU];
#pragma unroll
      for (int k = 0; k < kU; k++) {
        const int i = min(i0 + k * NT, n - 1);
        sid[k] = i < n_emit ? UX(x_bkt)[i] : -1 - p.unit_ilabel[u.tok_state[nb + i]];
        qv[k] = UX(x_q)[i];
      }
#pragma unroll
      for (int k = 0; k < kU; k++) {
        const int i = i0 + k * NT;
        if (i >= n) continue;
        const uint32_t bk = static_cast<uint32_t>(sid[k]) % H;
        T[i] = bk;
        if (i < n_emit) {
          const uint32_t m = 1u << (bk & 31u);
          const uint32_t old = __hip_atomic_fetch_or(&K[o1 + (bk >> 5)], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if ((old & m) != 0u) (void)__hip_atomic_fetch_or(&K[o2 + (bk >> 5)], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          (void)__hip_atomic_fetch_or(&K[qv[k] >> 5], 1u << (qv[k] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
  LdsSync();
  for (int i = n_emit + tid; i < n; i += NT) {   // a token of the closure in an occupied bucket shares it
    const uint32_t bk = T[i] & 0x1ffffu, m = 1u << (bk & 31u);
    if ((K[o1 + (bk >> 5)] & m) != 0u) (void)__hip_atomic_fetch_or(&K[o2 + (bk >> 5)], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  LdsSync();
  int n_b2;   // buckets with more than one token
  if (comb) {
    const int n_bits = BitmapPrefix(K, P, W1, sh);
    LdsSync();
    const int pre2 = static_cast<int>(P[o2]);
    n_b2 = n_bits - pre2;
    LdsSync();   // (every lane has read P[o2] = P2[0] before it is rewritten)
    for (int w = tid; w < Hw; w += NT) P2[w] -= static_cast<uint32_t>(pre2);   // (prefix counts of the shared-bucket bits alone)
    LdsSync();
  } else {
    n_b2 = BitmapPrefix(K + o2, P2, Hw, sh);
    LdsSync();
  }
  if (n_b2 > kLdsSlots / 2) return 2;
  // ---- 4. ranks, shared-bucket indices.  (The closure's tokens alone in their bucket keep the bucket id in a side array.)
  uint32_t tw[kU];   // (a lane's words: the bitmaps are read here, T is rewritten behind a barrier - the side array lies over them)
  int cb[kU];
#pragma unroll
  for (int k = 0; k < kU; k++) {
    tw[k] = 0u;
    cb[k] = -1;
    const int i = tid + k * NT;
    if (i >= n) continue;
    const uint32_t t0 = T[i], bk = t0 & 0x1ffffu, m = 1u << (bk & 31u);
    const uint32_t bits2 = K[o2 + (bk >> 5)];
    uint32_t word = 0u;
    if (i < n_emit) word = comb ? static_cast<uint32_t>(BitRank(K, P, UX(x_q)[i])) : t0 >> 17;
    if ((bits2 & m) != 0u) word |= ((P2[bk >> 5] + static_cast<uint32_t>(__popc(bits2 & (m - 1u)))) << 13) | (1u << 26);
    else if (i >= n_emit) cb[k] = static_cast<int>(bk);
    tw[k] = word;
  }
  LdsSync();   // (every lane has read the bitmaps and its T words)
  const LdsU32 Cb = K + 7168;          // [512] closure token -> bucket id + 1 if it is alone there, else 0
  const LdsU32 Cr = K + 7168 + 512;    // [512] closure token -> its rank
#pragma unroll
  for (int k = 0; k < kU; k++) {
    const int i = tid + k * NT;
    if (i >= n) continue;
    T[i] = tw[k];
    if (i >= n_emit) Cb[i - n_emit] = static_cast<uint32_t>(cb[k] + 1);
  }
  // ---- 5. the tokens of the shared buckets: counting sort by bucket index (counts K[0, n_b2), ranks as 16-bit words behind)
  const LdsU16 M16 = (LdsU16)(K + kLdsSlots / 2);
  if (n_b2 > 0) {
    for (int w = tid; w < n_b2; w += NT) K[w] = 0u;
    LdsSync();
    for (int i = tid; i < n_emit; i += NT) {
      const uint32_t w = T[i];
      if ((w >> 26) & 1u) (void)__hip_atomic_fetch_add(&K[(w >> 13) & 0x1fffu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    LdsSync();
    const int n_multi = LdsExScanInPlace(K, n_b2, sh);
    LdsSync();
    if (n_multi > 2 * (7168 - kLdsSlots / 2)) return 2;   // (the ranks' 16-bit words end where the closure's side arrays begin)
    for (int i = tid; i < n_emit; i += NT) {
      const uint32_t w = T[i];
      if ((w >> 26) & 1u)
        M16[__hip_atomic_fetch_add(&K[(w >> 13) & 0x1fffu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = static_cast<uint16_t>(w & 0x1fffu);
    }
    LdsSync();   // K[d] = end of bucket d's run = start of bucket d + 1's
    bool wide = false;
    for (int i = tid; i < n; i += NT) {
      const uint32_t w = T[i];
      if (((w >> 26) & 1u) == 0u) continue;
      const int d = static_cast<int>((w >> 13) & 0x1fffu);
      const int s0 = d > 0 ? static_cast<int>(K[d - 1]) : 0, s1 = static_cast<int>(K[d]);
      const uint32_t r = i < n_emit ? (w & 0x1fffu) : 0xFFFFFFFFu;   // (a token of the closure comes after all of them)
      uint32_t hm = 0xFFFFFFFFu;
      int inb = 0;
      for (int m = s0; m < s1; m++) {
        const uint32_t rr = M16[m];
        hm = rr < hm ? rr : hm;
        inb += rr < r ? 1 : 0;
      }
      wide |= inb > 255;
      T[i] = hm | (static_cast<uint32_t>(inb & 255) << 13);
    }
    if (wide) sh->flag = 1;
  } else {
    LdsSync();
  }
  LdsSync();
  if (Uni(sh->flag) != 0) return 2;   // (a bucket of more than 255 tokens)
  SubStamp(u, sh, 48);
  const LdsU32 kQ = K, kPop = K + 1024;
  const LdsF kNcost = (LdsF)(K + 2048);
  const LdsI kNlr = (LdsI)(K + 3072), kCode = (LdsI)(K + 4096);
  const LdsF kLw = (LdsF)(K + 5120);
  typedef __attribute__((address_space(3))) unsigned long long *LdsU64;
  const LdsU64 key64 = (LdsU64)(K + 6144 + ((reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void *)(K + 6144)) >> 2) & 1u));
  int my_rank = 0;   // closure token tid: its place among the closure's insertions
  if (n_new > 0) {
    // ---- 6. the closure: entries of tmp_epslist into T, the queue's order by counting, the tables, the walks
    for (int j = tid; j < eps_n; j += NT) {
      const int i = u.tmp_epslist[j] - nb;
      (void)__hip_atomic_fetch_or(&T[i], static_cast<uint32_t>(j + 1) << 21, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      kNcost[j] = j < eps_emit ? Dec(UX(x_c0e)[j]) : INFINITY;
      kNlr[j] = 0;
      if (j < eps_emit) {
        const uint32_t w = T[i];   // (its low 21 bits are final; the entry number lands above them)
        kQ[j] = ((w & 0x1fffu) << 8) | ((w >> 13) & 0xffu);
      }
    }
    for (int k = tid; k < n_new; k += NT) key64[k] = ~0ull;
    LdsSync();
    for (int j = tid; j < eps_emit; j += NT) {
      const uint32_t key = kQ[j];
      int rank = 0, j2 = 0;
      for (; j2 + 4 <= eps_emit; j2 += 4) {
        const uint32_t a0 = kQ[j2], a1 = kQ[j2 + 1], a2 = kQ[j2 + 2], a3 = kQ[j2 + 3];
        rank += (a0 < key ? 1 : 0) + (a1 < key ? 1 : 0) + (a2 < key ? 1 : 0) + (a3 < key ? 1 : 0);
      }
      for (; j2 < eps_emit; j2++) rank += kQ[j2] < key ? 1 : 0;
      kPop[j] = static_cast<uint32_t>(eps_emit - 1 - rank);
    }
    {
      constexpr int kLU = 4;
      for (int l0 = lb + tid; l0 < le; l0 += NT * kLU) {
        int srcs[kLU], dsts[kLU], arcs[kLU], prevs[kLU], nexts[kLU];
        float ws[kLU];
#pragma unroll
        for (int k = 0; k < kLU; k++) {
          const int l = min(l0 + k * NT, le - 1);
          srcs[k] = u.link_src[l];
          dsts[k] = u.link_dst[l];
          arcs[k] = u.link_arc[l];
          prevs[k] = l > lb ? u.link_src[l - 1] : -1;
          nexts[k] = l + 1 < le ? u.link_src[l + 1] : -1;
        }
#pragma unroll
        for (int k = 0; k < kLU; k++) ws[k] = __int_as_float(p.n_arcs[-1 - arcs[k]].z);
#pragma unroll
        for (int k = 0; k < kLU; k++) {
          const int l = l0 + k * NT;
          if (l >= le) continue;
          const int js = static_cast<int>(T[srcs[k] - nb] >> 21) - 1;
          int rng = 0;
          if (prevs[k] != srcs[k]) rng |= l - lb;
          if (nexts[k] != srcs[k]) rng |= (l - lb + 1) << 16;
          if (rng != 0) (void)__hip_atomic_fetch_or(&kNlr[js], rng, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          int code = -1;
          if (dsts[k] >= 0) {
            const int e = static_cast<int>(T[dsts[k] - nb] >> 21) - 1;
            if (e >= 0) code = e | ((dsts[k] >= ne_emit ? dsts[k] - ne_emit + 1 : 0) << 12);
            else if (dsts[k] >= ne_emit) code = 0x40000000 | (dsts[k] - ne_emit);
          }
          kCode[l - lb] = code;
          kLw[l - lb] = ws[k];
        }
      }
    }
    LdsSync();
    // (see OrderFrontierFast for why the first pop from which a path under the cutoff leads to a token is its insertion)
    for (int j = tid; j < eps_emit; j += NT) {
      const float c0 = kNcost[j];
      if (c0 > cutoff) continue;   // :779
      const unsigned long long pop = static_cast<unsigned long long>(kPop[j]) << 40;
      const int r = kNlr[j];
      for (int l = r & 0xffff; l < (r >> 16); l++) {
        int code = kCode[l];
        if (code < 0) continue;
        float tot = c0 + kLw[l];
        if (!(tot < cutoff)) continue;   // :794
        for (unsigned long long depth = 1;; depth++) {
          const bool leaf = (code & 0x40000000) != 0;
          const int k = leaf ? (code & 0x3fffffff) : (code >> 12) - 1;
          if (k < 0 || depth > 30) { sh->flag = 1; break; }
          (void)__hip_atomic_fetch_min(&key64[k], pop | (depth << 32) | static_cast<unsigned long long>(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (leaf) break;
          const int r2 = kNlr[code & 0xfff];
          int l2 = -1;
          for (int ll = r2 & 0xffff; ll < (r2 >> 16); ll++)
            if (kCode[ll] >= 0) { if (l2 >= 0) sh->flag = 1; l2 = ll; }
          if (l2 < 0) break;
          tot = tot + kLw[l2];
          if (!(tot < cutoff)) break;
          code = kCode[l2];
        }
      }
    }
    LdsSync();
    bool bad = false;
    if (tid < n_new) {
      const unsigned long long mine = key64[tid];
      bad = mine == ~0ull;
      const unsigned long long my_pop = mine >> 40, my_depth = (mine >> 32) & 0xffull, my_chain = mine & 0xffffffffull;
      for (int k2 = 0; k2 < n_new; k2++) {
        const unsigned long long o = key64[k2];
        my_rank += o < mine ? 1 : 0;
        if (k2 != tid && (o >> 40) == my_pop) {
          const unsigned long long od = (o >> 32) & 0xffull;
          if (od == my_depth && (my_depth >= 2 || o == mine)) bad = true;
          if (my_depth >= 2 && od >= 2 && (o & 0xffffffffull) != my_chain) bad = true;
        }
      }
    }
    if (bad) sh->flag = 1;
    LdsSync();
    if (Uni(sh->flag) != 0) return 2;   // (the queue's interpreter: OrderFrontierFast)
    if (u.phase_cycles != nullptr && tid == 0) sh->phase[64] += 1;
    // ---- 7. the closure's tokens: ranks; bucket mates among themselves (shared bucket: behind its tokens of the emitting pass)
    if (tid < n_new) Cr[tid] = static_cast<uint32_t>(n_emit + my_rank);
    LdsSync();
    if (tid < n_new) {
      const int i = n_emit + tid;
      const uint32_t w = T[i], r = Cr[tid], alone = Cb[tid];
      const uint32_t key = alone != 0u ? alone : 0x80000000u | (w & 0x1fffu);   // the bucket: its id, or the rank of its first token
      uint32_t minr = r;
      int before = 0;
      for (int k2 = 0; k2 < n_new; k2++) {
        const uint32_t a2 = Cb[k2];
        const uint32_t key2 = a2 != 0u ? a2 : 0x80000000u | (T[n_emit + k2] & 0x1fffu);
        if (key2 != key) continue;
        const uint32_t r2 = Cr[k2];
        before += r2 < r ? 1 : 0;
        minr = r2 < minr ? r2 : minr;
      }
      const int inb = (alone != 0u ? 0 : static_cast<int>((w >> 13) & 0xffu)) + before;
      if (inb > 255) sh->flag = 1;
      my_rank = static_cast<int>((alone != 0u ? minr : (w & 0x1fffu)) | (static_cast<uint32_t>(inb & 255) << 13));   // the token's (h, inb) word
    }
    LdsSync();   // (every closure lane has read the others' T words)
    if (Uni(sh->flag) != 0) return 2;
    if (tid < n_new) T[n_emit + tid] = (T[n_emit + tid] & 0xffe00000u) | static_cast<uint32_t>(my_rank);
  }
  SubStamp(u, sh, 50);
  // ---- 8. positions: tokens per first-of-bucket rank, one scan over the rank space, + the rank inside the bucket
  LdsSync();
  for (int w = tid; w < n; w += NT) K[w] = 0u;
  LdsSync();
  for (int i = tid; i < n; i += NT) (void)__hip_atomic_fetch_add(&K[T[i] & 0x1fffu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  LdsSync();
  (void)LdsExScanInPlace(K, n, sh);
  LdsSync();
  for (int i = tid; i < n; i += NT) {
    const uint32_t w = T[i];
    UX(x_pos)[i] = static_cast<int>(K[w & 0x1fffu]) + static_cast<int>((w >> 13) & 0xffu);
  }
  KhSync();
  SubStamp(u, sh, 51);
  if (u.phase_cycles != nullptr && tid == 0) {
    sh->phase[47] += 1; sh->phase[52] += eps_emit; sh->phase[53] += eps_n; sh->phase[54] += nl; sh->phase[55] += n_new; sh->phase[46] += n;
    sh->phase[66] += 1;
  }
  return 1;
}

// List positions (x_pos) of the frame under construction.  Params::exact_order == 2 forces the sort (tests).
__device__ bool OrderFrontier(const Utt &u, const Params &p, int nb, int fe, int lb, int le, float cutoff, Blk &sh) {
  const int n = fe - nb, n_new = fe - Uni(sh->x_ne_emit);
  const uint32_t H = Uni(sh->x_hsize);
  const bool fast = p.exact_order == 1 && n >= 1 && n <= kFastN && 2u * ((H + 31u) >> 5) <= static_cast<uint32_t>(kLdsSlots) && n_new <= kFastNew &&
                    Uni(sh->x_qbase) < 0x7fffffffu;
  long long t0 = 0;
  if (u.phase_cycles != nullptr && KH_TIDX == 0) {
    t0 = static_cast<long long>(__builtin_amdgcn_s_memtime());
    sh->phase[n <= 8160 ? 61 : (n <= 16384 ? 62 : 63)] += 1;
  }
  int rc = 2;
#ifndef KH_X_NO_LDS_TIER
  if (fast) rc = OrderFrontierLds(u, p, nb, fe, lb, le, cutoff, sh);
#endif
  if (fast && rc == 2) rc = OrderFrontierFast(u, p, nb, fe, lb, le, cutoff, sh);
  if (u.phase_cycles != nullptr && KH_TIDX == 0) {
    const long long t1 = static_cast<long long>(__builtin_amdgcn_s_memtime());
    if (rc == 1) sh->phase[59] += t1 - t0;
    t0 = t1;
  }
  if (rc == 2) {
#ifdef KH_X_NO_SORT
    if (KH_TIDX == 0) sh->status = 9;
    rc = 0;
#else
    rc = OrderFrontierSort(u, p, nb, fe, lb, le, cutoff, sh) ? 1 : 0;
#endif
    if (u.phase_cycles != nullptr && KH_TIDX == 0) {
      sh->phase[60] += static_cast<long long>(__builtin_amdgcn_s_memtime()) - t0;
      sh->phase[57] += 1;
    }
  }
  return rc == 1;
}

// ProcessEmitting :660-750 with the reference's running cutoff.  Tokens [b, e) with list positions x_pos.
// ONE sweep over the emitting arcs (round 5; rounds 1-4 read them twice).  The reference accepts arc k of the token at
// list position q iff tot_cost <= min(R[q], W[k]): R[q] = next_cutoff when the list walk reaches the token = min(estimate,
// tot_cost + adaptive_beam over the arcs of the tokens before it), W[k] = the same minimum over the token's own arcs
// before k.  (Rejected arcs never lower the running value: tot > cutoff implies tot + adaptive_beam > cutoff.)
//   * the sweep knows W[k] (a segmented prefix minimum inside the wave, carried over the token's 64-arc batches) but not
//     R[q]; it materialises a candidate iff tot_cost <= min(B, W[k]) for an upper bound B >= R[q], and leaves per token
//     M = min(tot_cost + adaptive_beam) and its arc count;
//   * B: the list is cut into 16 runs of positions; x_cb[j] = min(estimate, M of the tokens in runs < j seen so far) -
//     every contributor lies before every position of run j, so whatever subset has been published the value is an upper
//     bound of R there, and it tightens while the sweep runs (with the estimate alone the sweep would materialise every
//     arc the estimate admits: the densest part of the beam);
//   * ONE scan in list order - in LDS, over the position space - turns (M, count) into (R, ordinal of the token's first arc), stored by token;
//   * a sweep over the CANDIDATES (a quarter of the arcs, coalesced) drops those above R of their source token and gives
//     each its ordinal; pass 2 is the canonical one.
__device__ bool ProcessEmittingExact(const Utt &u, const Params &p, int frame, int b, int e, float *next_cutoff_out, int *cand_out,
                                     Blk &sh) {
  const int nb = Uni(sh->tok_end);  // first token of frame + 1
  const int tok_limit = min(u.tok_cap, nb + u.tok_frame_cap);
  const int n = e - b;
  if (KH_TIDX == 0) { sh->wl_n[0] = 0; sh->wl_n[1] = 0; sh->eps_n = 0; sh->erel_n = 0; }
  XS(27);
  StageScoreRow(u, p, sh, frame);
  Stamp(u, sh, 15);
  XS(24);
  const Cutoff c = GetCutoff<true>(u, p, b, e, sh);
  Stamp(u, sh, 0);
  XS(25);
  const int link_frame_b = Uni(sh->link_end);
  if (KH_TIDX == 0) {
    if (c.count > sh->max_tokens_frame) sh->max_tokens_frame = c.count;
    // PossiblyResizeHash(tok_cnt) :219-225
    const uint32_t new_sz = static_cast<uint32_t>(static_cast<float>(c.count) * p.hash_ratio);
    if (new_sz > sh->x_hsize) sh->x_hsize = new_sz;
  }
  const float inf = INFINITY;
  float cost_offset = 0.0f;
  float est = inf;
  if (c.best_tok >= 0) {
    cost_offset = -c.best_cost;  // :691
    const int32_t s = u.tok_state[c.best_tok];
    const float tot = c.best_cost;
    const int ab = s + 1, ae = ab + p.rec[s].x;
    for (int a = ab + KH_TIDX; a < ae; a += NT) {   // :692-704
      const KhInt4 arc = p.rec[a];
      const float w = __int_as_float(arc.z) + (cost_offset - LogLike(u, p, sh, frame, arc.x));
      const float new_weight = w + tot;
      est = fminf(est, new_weight + c.adaptive_beam);
    }
  }
  if (KH_TIDX == 0) {
    u.cost_offset[frame] = cost_offset;  // :710-711
    sh->work_cursor = b;
    sh->link_cursor = link_frame_b;
  }
  // the sweep's own copies of the pointers it uses (see ProcessEmitting: kept in scalar registers for its duration)
  Arr<const KhInt4> x_rec = p.rec;
  Arr<uint32_t> x_cost = u.tok_cost;
  Arr<int32_t> x_state = u.tok_state;
  GP(const float) x_ll = u.ll + static_cast<size_t>(frame) * u.ll_stride;
  int x_ll_cols = p.ll_cols;
  Arr<int32_t> xp_pos = UX(x_pos), xp_c = UX(x_c), xp_ord = UX(x_ord), xp_csid = UX(x_csid);
  Arr<uint32_t> xp_m = UX(x_m);
  Arr<int32_t> x_dst = u.link_dst, x_src = u.link_src, x_arc = u.link_arc;
  Arr<float> x_k = u.link_k, x_a = u.link_a;
  int x_keep_ac = p.keep_ac;
  KH_LAUNDER_X(x_rec.p); KH_LAUNDER_X(x_cost.p); KH_LAUNDER_X(x_state.p); KH_LAUNDER_X(x_ll); KH_LAUNDER_X(x_ll_cols);
  KH_LAUNDER_X(xp_pos.p); KH_LAUNDER_X(xp_c.p); KH_LAUNDER_X(xp_m.p); KH_LAUNDER_X(xp_ord.p); KH_LAUNDER_X(xp_csid.p);
  KH_LAUNDER_X(x_dst.p); KH_LAUNDER_X(x_src.p); KH_LAUNDER_X(x_arc.p); KH_LAUNDER_X(x_k.p); KH_LAUNDER_X(x_a.p);
  KH_LAUNDER_X(x_keep_ac);
  OwnerScan os = OwnerScanInit(sh);
  const float est0 = BlockMinF(est, sh);   // (its barrier publishes the cursors)
  XS(59);
  constexpr int kCB = 16;
  if (KH_TIDX < kCB) sh->x_cb[KH_TIDX] = Enc(est0);
  // run of positions a token belongs to: pos >> bshift, at most kCB runs
  const int bshift = n > kCB ? 32 - __clz(n - 1) - 4 : 0;
  const int limit = min(u.link_cap, link_frame_b + u.link_frame_cap);
  // Frames of up to kScanLds tokens: the sweep leaves a token's arc count at its list POSITION in LDS (K[pos]; the area
  // is idle during the sweep) and marks the few positions whose token has arcs under the bound in a bitmap behind the
  // counts (their M goes to x_m[pos]); the scan in list order then runs on LDS alone and the candidates read its results
  // by position.  Larger frames: (M, count) by token through memory, the scan in chunks of 8192 positions.
  constexpr int kScanLds = kLdsSlots - kLdsSlots / 32;
  const bool lds_scan = n <= kScanLds;
  // Frames of up to 2 * kLdsSlots tokens (nearly all of the rest; half of the kernel's work): the arc counts at their list
  // positions as 16-BIT words (the key area holds 16384 of them; a state's emitting fan-out fits - Params::max_emit), the
  // bitmap of the positions with a finite M in the top of the value area (above the score row, which the sweep is reading);
  // the scan then writes the ordinal of every position's first arc over BOTH areas (the score row is not read again) and
  // leaves the running cutoff as what it is - a step function with a handful of steps: the positions where it drops and
  // the values it drops to, as a short list in memory.  A candidate at or under the frame's FINAL cutoff is accepted
  // without looking (it is under every earlier value); the others - under one in a hundred - walk the list.
#ifndef KH_X_NO_MID_TIER
  constexpr int kMidBits = 2 * kLdsSlots / 32, kMidBitBase = kLdsSlots - kMidBits;
  const bool mid_scan = !lds_scan && n <= 2 * kLdsSlots && x_ll_cols <= kMidBitBase && p.max_emit < 65536 && p.exact_order == 1;
#else
  constexpr int kMidBits = 0, kMidBitBase = 0;
  const bool mid_scan = false;
#endif
  // Round 6: in the two LDS tiers the sweep leaves every candidate's place in the walk as (list position of its source
  // token) << 16 | (index of the arc among the token's arcs) in x_ord; the acceptance sweep turns it into the ordinal with
  // the scan's results alone - it used to read the candidate's source and arc and GATHER the source's position and state
  // (two gathers per candidate), and the candidates' source / arc words, not read again before pruning, go out with
  // non-temporal stores as in the canonical sweep.
#ifndef KH_X_NO_KEY_ORD
  const bool key_ord = (lds_scan || mid_scan) && p.max_emit < 65536;
#else
  const bool key_ord = false;
#endif
  const LdsU32 sK = (LdsU32)LdsKeys(sh);
  const LdsU32 sV = (LdsU32)LdsVals(sh);
  if (lds_scan)
    for (int w = n + static_cast<int>(KH_TIDX); w < n + ((n + 31) >> 5); w += NT) sK[w] = 0u;
  if (mid_scan) {
    for (int w = KH_TIDX; w < kMidBits; w += NT) sV[kMidBitBase + w] = 0u;
    if (KH_TIDX == 0) sh->x_nbp = 0;
  }
  LdsSync();
  const int lane = KH_TIDX & 63;
  long long my_arcs = 0;
  // inclusive minimum over the lanes that have the same owner (the lanes of a token are consecutive and `lo` is
  // non-decreasing): the wave scan with a segment test, on DPP
  auto seg_min_scan = [&](float m, int lo) -> float {
#ifndef KH_NO_DPP
#define KH_SEG_STEP(ctrl, rm) { const float nm = __int_as_float(DppMov<ctrl, rm>(0x7f800000, __float_as_int(m))); const int nlo = DppMov<ctrl, rm>(-2, lo); if (nlo == lo) m = fminf(m, nm); }
    KH_SEG_STEP(0x111, 0xf) KH_SEG_STEP(0x112, 0xf) KH_SEG_STEP(0x114, 0xf) KH_SEG_STEP(0x118, 0xf) KH_SEG_STEP(0x142, 0xa) KH_SEG_STEP(0x143, 0xc)
#undef KH_SEG_STEP
#else
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float nm = __shfl_up(m, o, 64);
      const int nlo = __shfl_up(lo, o, 64);
      if (lane >= o && nlo == lo) m = fminf(m, nm);
    }
#endif
    return m;
  };
  // ---- the sweep: candidates under min(bound of the token's run, the token's own earlier arcs); per token M and count.
  // Nearly every 64-arc batch holds no arc that lowers the cutoff (tot_cost + adaptive_beam under the bound takes an arc
  // better than the best token's best arc): then W[k] plays no part, the test is `tot_cost <= bound` and the batch costs
  // what it costs in the canonical sweep.  Only a batch with such an arc runs the segmented scans, and only such arcs
  // enter M (an arc at or above the bound of its token's run cannot lower R behind it either: M is exact wherever it matters).
  // A wave holds its next claim (cursor, cost, state, position requested under the current claim's batches).
  int base_next = WaveLdsFetchAdd(&sh->work_cursor, 64);
  uint32_t co_next = 0u;
  int st_next = 0, pos_next = 0;
  // (round 6, -DKH_X_CNT_PREFETCH: measured and NOT kept) hc_next: the emitting-arc count in the header of the next claim's
  // states, requested while the current claim's batches are in flight - the count is the first of a claim's dependent round
  // trips (state -> header -> arcs).  Same-box A/B 902 ms with it, 892 without: the extra register and the header reads of
  // tokens above the cutoff cost more than the round trip they hide next to 31 other waves.
  int hc_next = 0;
  if (base_next < e) {
    const int icn = min(base_next + lane, e - 1);
    co_next = LoadCostEnc(&x_cost[icn]);
    st_next = x_state[icn];
    pos_next = xp_pos[icn - b];
#ifdef KH_X_CNT_PREFETCH
    hc_next = x_rec[st_next].x;
#endif
  }
  for (;;) {
    const int base = base_next;
    if (base >= e) break;
    const int i = base + lane;
    const bool in_range = i < e;
    const uint32_t co = co_next;
    int st = st_next;
    const int pos = pos_next;
    const int hc = hc_next;
    const int blk = pos >> bshift;
    base_next = WaveLdsFetchAdd(&sh->work_cursor, 64);
    if (base_next < e) {
      const int icn = min(base_next + lane, e - 1);
      co_next = LoadCostEnc(&x_cost[icn]);
      st_next = x_state[icn];
      pos_next = xp_pos[icn - b];
    }
    bool hc_asked = false;
    KH_BOUND(1, st, 0, 0x7ffffff0);
    const bool need = in_range && Dec(co) <= c.cur_cutoff;
    int ab = 0, cnt = 0;
    if (need) {
      ab = st + 1;
#ifdef KH_X_CNT_PREFETCH
      cnt = hc;
#else
      cnt = x_rec[st].x;
#endif
    }
    const float bnd = Dec(__hip_atomic_load(&sh->x_cb[blk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    const int inc = WaveIncSum(cnt);
    const int loff = inc - cnt;
    const int total = WaveLast(inc);
    const int rel = ab - loff;   // arc index = rel(owner) + slot
    const int pk = (pos << 16) - loff;   // (position << 16 | arc's index within the token) = pk(owner) + slot
    const float cof = Dec(co);
    os.carry = -1;
    float acc = inf;   // min(tot_cost + adaptive_beam) over this token's arcs under the bound, in the batches before the current one
    for (int q0 = 0; q0 < total; q0 += 64) {
      const int q = q0 + lane;
      const bool valid = q < total;
      const int lo = OwnerLane(os, cnt, loff, q0, lane);
      const int o_rel = ShflI(rel, lo);
      const float o_co = ShflF(cof, lo);
      const float o_racc = ShflF(fminf(bnd, acc), lo);   // bound in front of the token, lowered by its arcs in earlier batches
      KhInt4 arc;
      arc.x = 0; arc.y = 0; arc.z = 0; arc.w = 0;
      float tot = inf, ac = 0.0f;
      const int ai = o_rel + q;
      if (valid) {
        arc = x_rec[ai];
        int32_t pdf = arc.x;
        KH_BOUND(7, pdf, 0, u.ll_stride);
        const float like = x_ll_cols > 0 ? sh.ll_row[pdf] : x_ll[pdf];
        ac = cost_offset - like;
        tot = o_co + ac + __int_as_float(arc.z);  // :726-730
      }
      const float m0 = tot + c.adaptive_beam;   // what this arc lowers next_cutoff to (:732-733)
      bool keep = valid && !(tot > o_racc) && tot == tot;   // :731 against an upper bound (a NaN candidate is dropped, as in the canonical rule)
      if (__ballot(valid && m0 < o_racc) != 0ull) {
        // some arc of the batch lowers the cutoff in front of the arcs behind it: the token's own earlier arcs count
        float m = (m0 == m0) ? m0 : inf;   // (a NaN never lowers it)
        m = seg_min_scan(m, lo);
#ifndef KH_NO_DPP
        const float pm = __int_as_float(DppMov<0x138, 0xf>(0x7f800000, __float_as_int(m)));   // wave_shr:1
        const int plo = DppMov<0x138, 0xf>(-2, lo);
        const float before = plo == lo ? pm : inf;
#else
        const float pm = __shfl_up(m, 1, 64);
        const int plo = __shfl_up(lo, 1, 64);
        const float before = (lane >= 1 && plo == lo) ? pm : inf;
#endif
        keep = keep && !(tot > before);
        const bool has = cnt > 0 && loff < q0 + 64 && loff + cnt > q0;
        const int tail = min(loff + cnt, q0 + 64) - 1 - q0;
        const float got = ShflF(m, has ? tail : 0);
        if (has) acc = fminf(acc, got);
      }
      const unsigned long long kb = __ballot(keep);
      if (kb != 0ull) {
        const int n_keep = __popcll(kb);
        const int at = WaveLdsFetchAdd(&sh->link_cursor, n_keep);
        if (at + n_keep > limit) {
          if (lane == 0) sh->status = (at + n_keep > u.link_cap) ? 2 : 3;
        } else {
          const int o_pk = key_ord ? ShflI(pk, lo) : 0;
          if (keep) {
            const int l = at + LanePrefixCount(kb);
            x_dst[l] = -2 - arc.w;
#ifndef KH_NO_NT
            if (key_ord) {
              __builtin_nontemporal_store(base + lo, &x_src[l]);
              __builtin_nontemporal_store(ai, &x_arc[l]);
              xp_ord[l - link_frame_b] = o_pk + q;
            } else {
              x_src[l] = base + lo;
              x_arc[l] = ai;
            }
            if (x_keep_ac) __builtin_nontemporal_store(ac, &x_a[l]);   // (not read before the next pruning visit)
#else
            x_src[l] = base + lo;
            x_arc[l] = ai;
            if (key_ord) xp_ord[l - link_frame_b] = o_pk + q;
            if (x_keep_ac) x_a[l] = ac;
#endif
            x_k[l] = tot;
#if !defined(KH_NO_NT) && !defined(KH_X_NO_NT_AUX)
            __builtin_nontemporal_store(arc.y, &xp_csid[l - link_frame_b]);   // the caller's id of the destination state (ArcPdfKernel); read once, by pass 2
#else
            xp_csid[l - link_frame_b] = arc.y;   // the caller's id of the destination state (ArcPdfKernel)
#endif
          }
        }
      }
#ifdef KH_X_CNT_PREFETCH
      if (!hc_asked) {   // behind the claim's FIRST batch: the next claim's states have landed (they were asked for before this batch's arcs)
        hc_asked = true;
        if (base_next < e) hc_next = x_rec[st_next].x;
      }
#endif
    }
#ifdef KH_X_CNT_PREFETCH
    if (!hc_asked && base_next < e) hc_next = x_rec[st_next].x;   // (a claim without arcs)
#endif
    if (in_range) {
      const uint32_t me = Enc(acc);   // (+inf for a token none of whose arcs came under the bound)
      if (lds_scan) {
        sK[pos] = static_cast<uint32_t>(cnt);
        if (me < kEncInf) {
          xp_m[pos] = me;
          (void)__hip_atomic_fetch_or(&sK[n + (pos >> 5)], 1u << (pos & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else if (mid_scan) {
        ((LdsU16)sK)[pos] = static_cast<uint16_t>(cnt);
        if (me < kEncInf) {
          xp_m[pos] = me;
          (void)__hip_atomic_fetch_or(&sV[kMidBitBase + (pos >> 5)], 1u << (pos & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {
        xp_m[i - b] = me;
        xp_c[i - b] = cnt;
      }
#ifndef KH_X_NO_RUN_BOUND
      if (me < kEncInf) {
#pragma nounroll
        for (int jj = blk + 1; jj < kCB; jj++) {
          const uint32_t old = __hip_atomic_fetch_min(&sh->x_cb[jj], me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (old <= me) break;   // (whoever put that value there carries it on to the runs behind)
        }
      }
#endif
    }
    if (lane == 0) my_arcs += total;
  }
  KhSync();
  XS(60);
  if (Uni(sh->status) != 0) return false;
  const int link_frame_e = Uni(sh->link_cursor);
  const TidRef tid;
  // ---- the scan in list order, in LDS over the POSITION space: every token puts (M, arc count) at its position, one
  // exclusive (min, sum) scan over the positions gives the running next_cutoff in front of every token (it starts at the
  // estimate) and the ordinal of its first arc, and the token takes them back: x_m = that cutoff, x_c = the ordinal of
  // its first arc minus the index of that arc (a candidate's ordinal = its arc index + x_c of its source).  No gather
  // through the inverse permutation, no workgroup scan per 1024 positions.  8192 positions at a time.
  uint32_t run_min = Enc(est0);
  int run_sum = 0;
  if (lds_scan) {
    // counts at K[0, n), the bitmap of positions with a finite M behind them; a lane owns eight consecutive positions
    const LdsU32 K = (LdsU32)LdsKeys(sh), V = (LdsU32)LdsVals(sh);   // (the score row is not read again in this frame)
    constexpr int kPer = kLdsSlots / NT;
    static_assert(kPer == 8, "a lane's positions are one byte of the bitmap");
    const int w0 = tid * kPer;
    uint32_t mm[kPer], lane_min = 0xFFFFFFFFu;
    int cc[kPer], lane_sum = 0;
    const uint32_t bits = w0 < n ? (K[n + (w0 >> 5)] >> (w0 & 31)) & 0xffu : 0u;
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const uint32_t kv = K[w0 + j];   // (unguarded: inside the area)
      cc[j] = w0 + j < n ? static_cast<int>(kv) : 0;
      mm[j] = 0xFFFFFFFFu;
      lane_sum += cc[j];
    }
    if (bits != 0u) {   // (rare: a token with arcs under the bound of its run)
#pragma unroll
      for (int j = 0; j < kPer; j++)
        if (((bits >> j) & 1u) != 0u) {
          mm[j] = xp_m[w0 + j];
          lane_min = mm[j] < lane_min ? mm[j] : lane_min;
        }
    }
    int ex_sum, tot_sum;
    uint32_t ex_min, tot_min;
    BlockExScanSumMin(lane_sum, lane_min, &ex_sum, &ex_min, &tot_sum, &tot_min, sh);   // (its barrier: every lane has read its counts and bits)
    uint32_t rm = run_min < ex_min ? run_min : ex_min;
    int rs = ex_sum;
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      if (w0 + j < n) { V[w0 + j] = rm; K[w0 + j] = static_cast<uint32_t>(rs); }
      rm = mm[j] < rm ? mm[j] : rm;
      rs += cc[j];
    }
    run_min = run_min < tot_min ? run_min : tot_min;
    run_sum = tot_sum;
    LdsSync();
  } else if (mid_scan) {
    // a lane owns sixteen consecutive positions: eight words of counts, sixteen bits of the bitmap
    const LdsU32 K = (LdsU32)LdsKeys(sh), V = (LdsU32)LdsVals(sh);
    constexpr int kPer = 2 * kLdsSlots / NT;
    static_assert(kPer == 16, "a lane's positions are half a word of the bitmap");
    const int w0 = tid * kPer;
    uint32_t mm[kPer], lane_min = 0xFFFFFFFFu;
    int cc[kPer], lane_sum = 0;
    const uint32_t bits = w0 < n ? (V[kMidBitBase + (w0 >> 5)] >> (w0 & 31)) & 0xffffu : 0u;
#pragma unroll
    for (int j = 0; j < kPer / 2; j++) {
      const uint32_t kv = K[(w0 >> 1) + j];   // (unguarded: inside the area)
      cc[2 * j] = w0 + 2 * j < n ? static_cast<int>(kv & 0xffffu) : 0;
      cc[2 * j + 1] = w0 + 2 * j + 1 < n ? static_cast<int>(kv >> 16) : 0;
      lane_sum += cc[2 * j] + cc[2 * j + 1];
    }
#pragma unroll
    for (int j = 0; j < kPer; j++) mm[j] = 0xFFFFFFFFu;
    if (bits != 0u) {   // (rare: a token with arcs under the bound of its run)
#pragma unroll
      for (int j = 0; j < kPer; j++)
        if (((bits >> j) & 1u) != 0u) {
          mm[j] = xp_m[w0 + j];
          lane_min = mm[j] < lane_min ? mm[j] : lane_min;
        }
    }
    int ex_sum, tot_sum;
    uint32_t ex_min, tot_min;
    BlockExScanSumMin(lane_sum, lane_min, &ex_sum, &ex_min, &tot_sum, &tot_min, sh);   // (its barrier: every lane has read its counts and bits)
    uint32_t rm = run_min < ex_min ? run_min : ex_min;
    int rs = ex_sum;
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const int w = w0 + j;
      if (w < n) { if (w < kLdsSlots) K[w] = static_cast<uint32_t>(rs); else V[w - kLdsSlots] = static_cast<uint32_t>(rs); }
      if (mm[j] < rm) {   // the running cutoff drops behind this position
        rm = mm[j];
        const int at = __hip_atomic_fetch_add(&sh->x_nbp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        xp_c[at] = w + 1;                               // (first position the new value holds for)
        UX(x_h)[at] = static_cast<int32_t>(rm);
      }
      rs += cc[j];
    }
    run_min = run_min < tot_min ? run_min : tot_min;
    run_sum = tot_sum;
    KhSync();   // (the steps of the running cutoff are in memory)
  } else {
    const LdsU32 K = (LdsU32)LdsKeys(sh), V = (LdsU32)LdsVals(sh);   // (the score row is not read again in this frame)
    constexpr int kTU = 4, kPer = kLdsSlots / NT;
    for (int p0 = 0; p0 < n; p0 += kLdsSlots) {
      const uint32_t W = static_cast<uint32_t>(min(kLdsSlots, n - p0));
      for (int i0 = tid; i0 < n; i0 += NT * kTU) {
        int ps[kTU], cs[kTU];
        uint32_t ms[kTU];
#pragma unroll
        for (int k = 0; k < kTU; k++) {
          const int i = min(i0 + k * NT, n - 1);
          ps[k] = xp_pos[i];
          ms[k] = xp_m[i];
          cs[k] = xp_c[i];
        }
#pragma unroll
        for (int k = 0; k < kTU; k++) {
          const uint32_t at = static_cast<uint32_t>(ps[k] - p0);
          if (i0 + k * NT < n && at < W) { K[at] = ms[k]; V[at] = static_cast<uint32_t>(cs[k]); }
        }
      }
      LdsSync();
      {
        uint32_t mm[kPer], lane_min = 0xFFFFFFFFu;
        int cc[kPer], lane_sum = 0;
#pragma unroll
        for (int j = 0; j < kPer; j++) {
          const uint32_t w = tid * kPer + j;
          const uint32_t kv = K[w], vv = V[w];   // (unguarded: the reads go out together)
          mm[j] = w < W ? kv : 0xFFFFFFFFu;
          cc[j] = w < W ? static_cast<int>(vv) : 0;
          lane_min = mm[j] < lane_min ? mm[j] : lane_min;
          lane_sum += cc[j];
        }
        int ex_sum, tot_sum;
        uint32_t ex_min, tot_min;
        BlockExScanSumMin(lane_sum, lane_min, &ex_sum, &ex_min, &tot_sum, &tot_min, sh);
        uint32_t rm = run_min < ex_min ? run_min : ex_min;
        int rs = run_sum + ex_sum;
#pragma unroll
        for (int j = 0; j < kPer; j++) {
          const uint32_t w = tid * kPer + j;
          if (w < W) { K[w] = rm; V[w] = static_cast<uint32_t>(rs); }
          rm = mm[j] < rm ? mm[j] : rm;
          rs += cc[j];
        }
        run_min = run_min < tot_min ? run_min : tot_min;
        run_sum += tot_sum;
      }
      LdsSync();
      for (int i0 = tid; i0 < n; i0 += NT * kTU) {
        int ps[kTU], sts[kTU];
#pragma unroll
        for (int k = 0; k < kTU; k++) {
          const int i = min(i0 + k * NT, n - 1);
          ps[k] = xp_pos[i];
          sts[k] = x_state[b + i];
        }
#pragma unroll
        for (int k = 0; k < kTU; k++) {
          const uint32_t at = static_cast<uint32_t>(ps[k] - p0);
          if (i0 + k * NT < n && at < W) {
            xp_m[i0 + k * NT] = K[at];
            xp_c[i0 + k * NT] = static_cast<int>(V[at]) - (sts[k] + 1);
          }
        }
      }
      LdsSync();
    }
  }
  const float next_cutoff = Dec(run_min);   // the value the running cutoff ends at
  XS(61);
  if (tid == 0) {
    sh->x_qbase = static_cast<uint32_t>(run_sum);
    u.femit_b[frame] = link_frame_b;
    u.femit_e[frame] = link_frame_e;
    sh->link_end = link_frame_e;
    sh->front_b = nb;
  }
  if (lds_scan) LdsSync(); else KhSync();   // (the scan through memory: its results by token have landed)
  const int n_bp = mid_scan ? Uni(sh->x_nbp) : 0;
  const uint32_t r_final = run_min, r_start = Enc(est0);
  // ---- the candidates against the running cutoff in front of their source token (:731); ordinals
  int n_acc = 0;
  {
    constexpr int kOU = 4;
    for (int l0 = link_frame_b + tid; l0 < link_frame_e; l0 += NT * kOU) {
      float ks[kOU];
      int srcs[kOU], ords[kOU], as[kOU];
      uint32_t rs[kOU];
      if (key_ord) {   // the candidate's (position, arc index) as the sweep left it: no gather
        const LdsU32 K = (LdsU32)LdsKeys(sh), V = (LdsU32)LdsVals(sh);
#pragma unroll
        for (int k = 0; k < kOU; k++) {
          const int l = min(l0 + k * NT, link_frame_e - 1);
          ks[k] = x_k[l];
          ords[k] = xp_ord[l - link_frame_b];
        }
#pragma unroll
        for (int k = 0; k < kOU; k++) {
          const int ps = static_cast<int>(static_cast<uint32_t>(ords[k]) >> 16);
          ords[k] &= 0xffff;
          if (lds_scan) {
            rs[k] = V[ps];
            as[k] = static_cast<int>(K[ps]);
          } else {
            as[k] = static_cast<int>(ps < kLdsSlots ? K[ps] : V[ps - kLdsSlots]);
            rs[k] = r_final;
            if (ks[k] > Dec(r_final)) {   // (rare) above the final cutoff: the value that held in front of its source token
              uint32_t r = r_start;
              for (int q = 0; q < n_bp; q++) {
                const uint32_t v = static_cast<uint32_t>(UX(x_h)[q]);
                if (xp_c[q] <= ps && v < r) r = v;
              }
              rs[k] = r;
            }
          }
        }
      } else {
#pragma unroll
      for (int k = 0; k < kOU; k++) {
        const int l = min(l0 + k * NT, link_frame_e - 1);
        ks[k] = x_k[l];
        srcs[k] = x_src[l];
        ords[k] = x_arc[l];
      }
      if (lds_scan) {   // the scan's results by list position: V = the running cutoff, K = the ordinal of the token's first arc
        const LdsU32 K = (LdsU32)LdsKeys(sh), V = (LdsU32)LdsVals(sh);
        int ps[kOU], sts[kOU];
#pragma unroll
        for (int k = 0; k < kOU; k++) {
          ps[k] = xp_pos[srcs[k] - b];
          sts[k] = x_state[srcs[k]];
        }
#pragma unroll
        for (int k = 0; k < kOU; k++) {
          rs[k] = V[ps[k]];
          as[k] = static_cast<int>(K[ps[k]]) - (sts[k] + 1);
        }
      } else if (mid_scan) {   // the ordinal of the token's first arc by list position, over both areas; the running cutoff from its steps
        const LdsU32 K = (LdsU32)LdsKeys(sh), V = (LdsU32)LdsVals(sh);
        int ps[kOU], sts[kOU];
#pragma unroll
        for (int k = 0; k < kOU; k++) {
          ps[k] = xp_pos[srcs[k] - b];
          sts[k] = x_state[srcs[k]];
        }
#pragma unroll
        for (int k = 0; k < kOU; k++) {
          const uint32_t base = ps[k] < kLdsSlots ? K[ps[k]] : V[ps[k] - kLdsSlots];
          as[k] = static_cast<int>(base) - (sts[k] + 1);
          rs[k] = r_final;
          if (ks[k] > Dec(r_final)) {   // (rare) above the final cutoff: the value that held in front of its source token
            uint32_t r = r_start;
            for (int q = 0; q < n_bp; q++) {
              const uint32_t v = static_cast<uint32_t>(UX(x_h)[q]);
              if (xp_c[q] <= ps[k] && v < r) r = v;
            }
            rs[k] = r;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < kOU; k++) {
          rs[k] = xp_m[srcs[k] - b];
          as[k] = xp_c[srcs[k] - b];
        }
      }
      }
#pragma unroll
      for (int k = 0; k < kOU; k++) {
        const int l = l0 + k * NT;
        if (l >= link_frame_e) continue;
        if (ks[k] > Dec(rs[k])) x_dst[l] = -1;   // the reference's `continue`: nothing was made of this arc
        else n_acc++;
#if !defined(KH_NO_NT) && !defined(KH_X_NO_NT_AUX)
        __builtin_nontemporal_store(ords[k] + as[k], &xp_ord[l - link_frame_b]);   // (read once, by pass 2)
#else
        xp_ord[l - link_frame_b] = ords[k] + as[k];
#endif
      }
    }
  }
  n_acc = static_cast<int>(BlockSumLL(n_acc, sh));   // (its barrier: the dead candidates and the ordinals are in place)
  if (u.phase_cycles != nullptr && tid == 0) sh->phase[58] += n_acc;
  XS(62);
  Stamp(u, sh, 1);
  // ---- pass 2: FindOrAddToken + minimum cost in the LDS token table, as in the canonical sweep (every live candidate
  // has been accepted: no cutoff test)
  // (the insertion keys inside pass 2 when the ordinals fit its 19 bits - every frame of the benchmark; else the sweep below)
#ifndef KH_X_NO_FUSED_KEYS
  const bool fuse_ord = run_sum <= 0x7FFFF && p.exact_order == 1;   // (exact_order 2, KH_DECODER_ORDER_SORT: round 4's constructions throughout - the tests)
#else
  const bool fuse_ord = false;
#endif
  if (!EmitPass2<true>(u, sh, nb, tok_limit, link_frame_b, link_frame_e, inf, n_acc, fuse_ord)) return false;
  XS(63);
  // ---- the insertion key of a new token = the smallest ordinal among its candidates (the arc that made the reference
  // call HashList::Insert for it); its cost before the closure
  if (!fuse_ord) {
    const int n_new = Uni(sh->tok_end) - nb;
    auto qtab = LdsKeys(sh);
    const bool in_lds = n_new <= kLdsSlots;
    const bool sid_lds = 2 * n_new <= kLdsSlots;
    auto stab = LdsKeys(sh) + kLdsSlots / 2;
    for (int i = tid; i < n_new; i += NT) {
      if (in_lds) qtab[i] = 0xFFFFFFFFu; else UX(x_q)[i] = 0xFFFFFFFFu;
    }
    // (pass 2 ended behind a barrier that waited for its stores: the links' destinations are in memory; only a table in
    // memory needs another such barrier)
    if (in_lds) LdsSync(); else KhSync();
    {
      constexpr int kOU = 4;   // (a lane's loads of four trips in flight together)
      for (int l0 = link_frame_b + tid; l0 < link_frame_e; l0 += NT * kOU) {
        int dsts[kOU], sids[kOU];
        uint32_t ords[kOU];
#pragma unroll
        for (int k = 0; k < kOU; k++) {
          const int l = min(l0 + k * NT, link_frame_e - 1);
          dsts[k] = u.link_dst[l];
          ords[k] = static_cast<uint32_t>(UX(x_ord)[l - link_frame_b]);
          sids[k] = UX(x_csid)[l - link_frame_b];
        }
#pragma unroll
        for (int k = 0; k < kOU; k++) {
          if (l0 + k * NT >= link_frame_e || dsts[k] < 0) continue;   // (dst < 0: rejected by the running cutoff)
          if (in_lds) (void)__hip_atomic_fetch_min(&qtab[dsts[k] - nb], ords[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else (void)__hip_atomic_fetch_min(&UX(x_q)[dsts[k] - nb], ords[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // the new token's state id (every candidate of a token carries the same one): through LDS, so that it reaches
          // memory as one coalesced sweep (a scattered 4-byte store per candidate was 0.5 MB of partial writes per frame)
          if (sid_lds) stab[dsts[k] - nb] = static_cast<uint32_t>(sids[k]);
          else UX(x_bkt)[dsts[k] - nb] = sids[k];
        }
      }
    }
    LdsSync();
    if (in_lds)
      for (int i = tid; i < n_new; i += NT) {
        UX(x_q)[i] = qtab[i];
        if (sid_lds) UX(x_bkt)[i] = static_cast<int32_t>(stab[i]);
      }
  }
  // the arcs visited, for the utterance's counter: one LDS add per wave (lane 0 holds the wave's count), no block reduction;
  // the insertion keys and state ids are read by the list order, several store-draining barriers from here
  if ((tid & 63) == 0 && my_arcs != 0) (void)__hip_atomic_fetch_add(&sh->arcs_expanded, my_arcs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (tid == 0) {
    sh->wl_n[0] = sh->eps_n;
    sh->x_eps_emit = sh->eps_n;
    sh->x_ne_emit = sh->tok_end;
  }
  LdsSync();
  XS(58);
  Stamp(u, sh, 2);
  *next_cutoff_out = next_cutoff;
  *cand_out = link_frame_e - link_frame_b;
  return Uni(sh->status) == 0;
}


// ---------------------------------------------------------------- pruning
// PruneForwardLinks walks LINKS, not tokens: one lane per link slot (coalesced,
// branch-free loads, PU slots per lane in flight), the per-token minimum of
// link_extra_cost (:309-323) is an atomicMin on the order-preserving image Enc().

// Accumulator of PruneForwardLinks (per token of the frame: Enc(min link_extra_cost)): in LDS
// when the frame has at most kLdsSlots tokens (nearly always: every global atomic is a DRAM
// read-modify-write on this pool), else in the slot's global scratch array.
struct Acc {
  bool lds;
  __attribute__((address_space(3))) uint32_t *l;
  Arr<uint32_t> g;
  __device__ __forceinline__ void Set(int i, uint32_t v) const {
    if (lds) l[i] = v; else g[i] = v;
  }
  __device__ __forceinline__ uint32_t Get(int i) const {
    return lds ? l[i] : LoadCostEnc(&g[i]);
  }
  __device__ __forceinline__ void Min(int i, uint32_t v) const {
    if (lds) (void)__hip_atomic_fetch_min(&l[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else (void)__hip_atomic_fetch_min(&g[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

// One pass over the link slots [lo, hi).  link_extra_cost (:309-311) of every
// live link with the extra_costs currently stored; kExcise: links over the
// lattice beam are excised (:315); kAccum: the others are min-ed into
// acc[src - b].  Returns 2 if this lane excised a link.
template <bool kEps, bool kAccum, bool kExcise, bool kList = false>
__device__ __forceinline__ int PruneLinkPass(const Utt &u, int lo, int hi, int b, float lb, const Acc &acc, bool fresh,
                                             __attribute__((address_space(3))) int *list_n = nullptr) {
  int flags = 0;
  for (int base = lo + KH_TIDX; base < hi; base += NT * PU) {
    int l[PU], dst[PU], src[PU];
    float kk[PU];
#pragma unroll
    for (int k = 0; k < PU; k++) {
      l[k] = min(base + k * NT, hi - 1);  // clamped: the tail repeats the last slot
      dst[k] = u.link_dst[l[k]];
      src[k] = u.link_src[l[k]];
      kk[k] = u.link_k[l[k]];
    }
    float ex[PU];
#pragma unroll
    for (int k = 0; k < PU; k++) {
      ex[k] = 0.0f;
      if (dst[k] < 0) continue;  // excised slot: no gathers
      // fresh (the frame's first visit): link_k of an emitting link still holds the candidate's tot_cost
      if (!kEps && fresh) kk[k] = kk[k] - Dec(LoadCostEnc(&u.tok_cost[dst[k]]));
      ex[k] = LoadExtra(&u.tok_extra[dst[k]]);
    }
#pragma unroll
    for (int k = 0; k < PU; k++) {
      if (base + k * NT >= hi || dst[k] < 0) continue;
      if (kList) {
        // list (once) the tokens that own live epsilon links; mark the tokens such links lead to
        // (bit 1: a change of THEIR extra_cost is what makes another sweep necessary)
        if ((__hip_atomic_fetch_or(&u.tmp_dirty[src[k] - b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) == 0)
          u.tmp_work0[__hip_atomic_fetch_add(list_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = src[k];
        (void)__hip_atomic_fetch_or(&u.tmp_dirty[dst[k] - b], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (kEps && kExcise) u.tmp_dirty[dst[k] - b] = 0;  // last pass over the epsilon links: leave tmp_dirty all zero
      // :309-311; the parenthesis is link_k
      float lec = ex[k] + kk[k];
      if (lec > lb) {  // :315 excise
        if (kExcise) { u.link_dst[l[k]] = -1; flags |= 2; }
        continue;
      }
      if (!kEps && fresh) u.link_k[l[k]] = kk[k];
      if (kAccum) {
        if (lec < 0.0f) lec = 0.0f;  // :319-320
        acc.Min(src[k] - b, Enc(lec));
      }
    }
  }
  return flags;
}

// PruneForwardLinks :273-344 (canonical rule P: exact fixed point, then excise)
// for frame f = tokens [b, e), emitting links [mb, me), epsilon links [nb, ne).
// final_frame: PruneForwardLinksFinal :349-431.  [tb, te): tokens of frame f + 1
// to run PruneTokensForFrame :450-469 on in the same sweep (tb == te: none).
__device__ void PruneForwardLinks(const Utt &u, const Params &p, int b, int e, int mb, int me, int nb, int ne,
                                  float delta, bool final_frame, bool have_final, float final_best_cost,
                                  int tb, int te, bool fresh, bool *extra_costs_changed, bool *links_pruned, Blk &sh) {
  const float inf = INFINITY;
  const float lb = p.lattice_beam;
  if (u.phase_cycles != nullptr && KH_TIDX == 0) { sh->phase[10] += 1; sh->phase[11] += e - b; }
  long long tp = 0;
  const bool prof = u.phase_cycles != nullptr && KH_TIDX == 0 && e - b > NT;
  if (prof) tp = static_cast<long long>(__builtin_amdgcn_s_memtime());
#define KH_PRUNE_STAMP(k) do { if (prof) { const long long now_ = static_cast<long long>(__builtin_amdgcn_s_memtime()); sh->phase[k] += now_ - tp; tp = now_; } } while (0)
  // P0 (tokens): start the accumulator of the emitting links.  (tmp_acc1, the one of
  // the epsilon links, is +inf for every token outside this function.)
  if (KH_TIDX == 0) { sh->wl_n[0] = 0; sh->pr_moved = 0; }
  Acc acc0, acc1;
  acc0.lds = e - b <= kLdsSlots;  // (uniform) the LDS of the emitting pass's token table is idle here
  acc0.l = LdsVals(sh);
  acc0.g = u.tmp_acc0;
  acc1.lds = false;
  acc1.l = nullptr;
  acc1.g = u.tmp_acc1;
  for (int base = b + KH_TIDX; base < e; base += NT * PU) {
    int i[PU], st[PU];
    uint32_t co[PU];
#pragma unroll
    for (int k = 0; k < PU; k++) {
      i[k] = min(base + k * NT, e - 1);
      st[k] = final_frame ? u.tok_state[i[k]] : 0;
      co[k] = final_frame ? LoadCostEnc(&u.tok_cost[i[k]]) : 0u;
    }
    float fc[PU];
#pragma unroll
    for (int k = 0; k < PU; k++) fc[k] = (final_frame && have_final && st[k] >= 0) ? __int_as_float(p.rec[st[k]].w) : 0.0f;
#pragma unroll
    for (int k = 0; k < PU; k++) {
      if (base + k * NT >= e) continue;
      float base_v = inf;
      if (final_frame) base_v = Dec(co[k]) + fc[k] - final_best_cost;  // :385
      acc0.Set(i[k] - b, Enc(base_v));
    }
  }
  // PruneTokensForFrame(f + 1): its extra_costs are final, nothing below reads its states
  for (int i = tb + KH_TIDX; i < te; i += NT)
    if (u.tok_state[i] >= 0 && LoadExtra(&u.tok_extra[i]) == inf) u.tok_state[i] = -1;
  KhSync();
  KH_PRUNE_STAMP(20);
  // P1 (emitting links): a link to the NEXT frame sees final extra_costs there, so
  // its link_extra_cost - hence whether it is excised - is final at first sight.
  int flags = PruneLinkPass<false, true, true>(u, mb, me, b, lb, acc0, fresh);
  KhSync();
  KH_PRUNE_STAMP(21);
  // Epsilon links stay inside the frame (a DAG): the exact fixed point of
  //   extra[t] = min(emitting part, min over t's epsilon links of extra[dst] + const)
  // is unique, so the order of evaluation is free.  T0 gives every token its emitting part
  // (final for the tokens without epsilon links); then each round = one sweep over the
  // epsilon links with the current extra_costs + one over the tokens that own them (listed
  // by the first sweep).  A further round is needed only if a token that some epsilon link
  // LEADS TO has changed (marked by the first sweep): rounds = depth of the DAG, no
  // verification round and no round on the stale extra_costs of the previous visit.
  int n_moved = 0;
  auto settle = [&](int i, uint32_t a0, uint32_t a1, int st, float old, float entry, bool &changed) {
    if (st < 0) return;
    float v = Dec(a1 < a0 ? a1 : a0);
    if (final_frame && v > lb) v = inf;  // :416-417
    if (!(v == old)) {
      StoreExtra(&u.tok_extra[i], v);
      changed = true;
    }
    // :334 on the converged values: net count of tokens whose extra_cost moved by more than delta
    n_moved += (fabsf(v - entry) > delta ? 1 : 0) - (fabsf(old - entry) > delta ? 1 : 0);
    if (a1 != kEncInf) u.tmp_acc1[i - b] = kEncInf;
  };
  for (int i = b + KH_TIDX; i < e; i += NT) {  // T0 (tmp_acc1 is +inf for every token here)
    const uint32_t a0 = acc0.Get(i - b);
    const int st = u.tok_state[i];
    const float old = LoadExtra(&u.tok_extra[i]);
    u.tmp_f0[i - b] = old;
    bool changed = false;
    settle(i, a0, kEncInf, st, old, old, changed);
  }
  if (u.phase_cycles != nullptr && KH_TIDX == 0) sh->phase[14] += 1;
  if (ne > nb) {
    for (int iter = 0;; iter++) {
      WM(42, iter);
      KhSync();  // the extra_costs of the previous token sweep are in place
      if (iter == 0) PruneLinkPass<true, true, false, true>(u, nb, ne, b, lb, acc1, false, &sh->wl_n[0]);
      else PruneLinkPass<true, true, false>(u, nb, ne, b, lb, acc1, false);
      KhSync();
      bool again = false;
      const int n_list = Uni(sh->wl_n[0]);
      for (int q = KH_TIDX; q < n_list; q += NT) {
        const int i = u.tmp_work0[q];
        const uint32_t a1 = LoadCostEnc(&u.tmp_acc1[i - b]), a0 = acc0.Get(i - b);
        const int st = u.tok_state[i];
        const float old = LoadExtra(&u.tok_extra[i]), entry = u.tmp_f0[i - b];
        bool changed = false;
        settle(i, a0, a1, st, old, entry, changed);
        // (L2 read: the marks were set by atomics, which do not refresh this CU's L1)
        if (changed && (__hip_atomic_load(&u.tmp_dirty[i - b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) != 0) again = true;
      }
      if (u.phase_cycles != nullptr && KH_TIDX == 0) sh->phase[14] += 1;
      if (!BlockAny(again, sh)) break;
      if (iter > e - b + 8) { if (KH_TIDX == 0) sh->status = 10; break; }   // (not a DAG: see PruneFrameLds)
    }
    // bit 0 of the listed owners (bit 1 of the destinations is cleared by the excise pass below)
    const int n_list = Uni(sh->wl_n[0]);
    for (int q = KH_TIDX; q < n_list; q += NT) u.tmp_dirty[u.tmp_work0[q] - b] = 0;
  }
  if (n_moved != 0) __hip_atomic_fetch_add(&sh->pr_moved, n_moved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  KH_PRUNE_STAMP(22);
  if (ne > nb) flags |= PruneLinkPass<true, false, true>(u, nb, ne, b, lb, acc1, false);
  int all = BlockOr(flags, sh);
  if (Uni(sh->pr_moved) > 0) all |= 1;
  KH_PRUNE_STAMP(23);
#undef KH_PRUNE_STAMP
  *extra_costs_changed = (all & 1) != 0;
  *links_pruned = (all & 2) != 0;
}

// PruneForwardLinks :273-344 (+ PruneTokensForFrame(f + 1) when prune_toks_f1) for a frame of
// at most kPruneLdsTok tokens whose successor frame is no larger - nearly every visit of the
// backward pruning.  The general routine above is a chain of ~10 barrier-separated phases,
// each with its own global round trip (gathers of cost[src], cost[dst], extra[dst] per link
// and per iteration).  Here everything the visit needs is read ONCE, coalesced, into LDS (the
// idle 64 KB of the emitting pass's token table: costs of f, costs and extra_costs of f + 1,
// the two accumulators, the extra_costs being computed), every gather and the whole fixed
// point run on LDS, and the results are written back: same float operations, same unique
// fixed point.
constexpr int kPruneLdsTok = kLdsSlots / 2;  // 4096
__device__ void PruneFrameLds(const Utt &u, const Params &p, int b, int e, int mb, int me, int nb, int ne, int b1, int e1,
                              bool prune_toks_f1, bool fresh, float delta, bool *extra_costs_changed, bool *links_pruned, Blk &sh) {
  const float inf = INFINITY, lb = p.lattice_beam;
  const int t = KH_TIDX;
  auto ra = LdsKeys(sh);  // 8192 words
  auto rb = LdsVals(sh);  // 8192 words
  auto s_acc0 = ra + kPruneLdsTok;                                                                  // Enc(min) over emitting links
  auto s_nc = reinterpret_cast<__attribute__((address_space(3))) float *>(rb);                    // phase 1, first visit: cost of f + 1's tokens
  auto s_nx = reinterpret_cast<__attribute__((address_space(3))) float *>(rb + kPruneLdsTok);     // phase 1: extra_cost of f + 1's tokens
  auto s_acc1 = rb;                                                                                 // phase 2: Enc(min) over epsilon links
  auto s_x = reinterpret_cast<__attribute__((address_space(3))) float *>(rb + kPruneLdsTok);      // phase 2: extra_cost being computed
  if (u.phase_cycles != nullptr && t == 0) { sh->phase[10] += 1; sh->phase[11] += e - b; sh->phase[14] += 1; }
  // ---- everything the visit reads, in one round trip
  for (int i = t; i < e - b; i += NT) s_acc0[i] = kEncInf;
  for (int i = t; i < e1 - b1; i += NT) {
    const float nx = LoadExtra(&u.tok_extra[b1 + i]);
    if (fresh) s_nc[i] = Dec(LoadCostEnc(&u.tok_cost[b1 + i]));
    s_nx[i] = nx;
    // PruneTokensForFrame(f + 1) :450-469: its extra_costs are final
    if (prune_toks_f1 && nx == inf && u.tok_state[b1 + i] >= 0) u.tok_state[b1 + i] = -1;
  }
  // the first epsilon link of every lane stays in registers over the iterations
  int n_dst = -1, n_src = 0;
  float n_a = 0.0f;
  if (nb + t < ne) { n_dst = u.link_dst[nb + t]; n_src = u.link_src[nb + t]; n_a = u.link_k[nb + t]; }
  // ... and the first emitting link is fetched in the same round trip as the tokens
  int m_dst = -1, m_src = 0;
  float m_k = 0.0f;
  if (mb + t < me) { m_dst = u.link_dst[mb + t]; m_src = u.link_src[mb + t]; m_k = u.link_k[mb + t]; }
  WM(24, ne - nb);
  KhSync();
  WM(25, me - mb);
  // ---- emitting links (to frame f + 1, whose extra_costs are final): :309-323
  int flags = 0;
  for (int l = mb + t; l < me; l += NT) {
    const bool first = l == mb + t;
    const int dst = first ? m_dst : u.link_dst[l];
    if (dst < 0) continue;
    const int src = first ? m_src : u.link_src[l];
    float kk = first ? m_k : u.link_k[l];
    if (fresh) kk = kk - s_nc[dst - b1];  // first visit: link_k still holds the candidate's tot_cost (:309-311's parenthesis = tot_cost - cost[dst])
    float lec = s_nx[dst - b1] + kk;
    if (lec > lb) {
      u.link_dst[l] = -1;
      flags |= 2;
    } else {
      if (fresh) u.link_k[l] = kk;
      if (lec < 0.0f) lec = 0.0f;
      __hip_atomic_fetch_min(&s_acc0[src - b], Enc(lec), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  // what the write-back compares with: fetched now, used behind the epsilon fixed point
  constexpr int kTPL = kPruneLdsTok / NT;
  float old_x[kTPL];
  int old_st[kTPL];
#pragma unroll
  for (int k = 0; k < kTPL; k++) {
    const int i = t + k * NT;
    old_x[k] = 0.0f;
    old_st[k] = -1;
    if (i < e - b) { old_x[k] = LoadExtra(&u.tok_extra[b + i]); old_st[k] = u.tok_state[b + i]; }
  }
  WM(26, 0);
  LdsSync();  // (s_nc / s_nx are dead from here on: their LDS becomes s_acc1 / s_x)
  for (int i = t; i < e - b; i += NT) {
    s_x[i] = Dec(s_acc0[i]);
    s_acc1[i] = kEncInf;
  }
  // ---- epsilon links (inside the frame): iterate to the fixed point
  if (ne > nb) {
    for (int round_no = 0;; round_no++) {
      WM(27, round_no);
#ifdef KH_SERVE_MARKERS
      if (round_no == 5000 && g_wave_mark != nullptr) {   // (a frame's links form a DAG of depth <= its tokens: this round is never reached)
        // what is wrong with the frame's epsilon links?  words 16.. of the stream's debug block
        int32_t *dbg = g_wave_mark + blockIdx.x * 64 + 16;
        if (t == 0) { for (int k = 0; k < 8; k++) sh->phase[k] = 0; }
        KhSync();
        for (int l = nb + t; l < ne; l += NT) {
          const int dst = u.link_dst[l], src = u.link_src[l];
          const float kv = u.link_k[l];
          if (dst < 0) { __hip_atomic_fetch_add(&sh->phase[4], 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); continue; }
          int cat = -1;
          if (dst < b || dst >= e) cat = 0;
          else if (src < b || src >= e) cat = 1;
          else if (!(kv >= 0.0f)) cat = 2;            // negative or not a number
          else if (!(kv < INFINITY)) cat = 3;
          if (cat >= 0) {
            __hip_atomic_fetch_add(&sh->phase[cat], 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int32_t *r = dbg + 8 + 4 * cat;
            __hip_atomic_store(r + 0, l - nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(r + 1, dst - b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(r + 2, src - b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(r + 3, __float_as_int(kv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          if (dst >= b && dst < e && dst == src) __hip_atomic_fetch_add(&sh->phase[5], 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          // a link that leads to a token created before its source: the closure appends a token when it first reaches its
          // state, so such a link exists in a DAG too - counted, with one sample
          if (dst >= b && dst < e && dst < src) {
            __hip_atomic_fetch_add(&sh->phase[6], 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int32_t *r = dbg + 24;
            __hip_atomic_store(r + 0, l - nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(r + 1, dst - b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(r + 2, src - b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(r + 3, __float_as_int(kv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        // which tokens keep changing?  (their count this round, one sample with its values)
        KhSync();
        if (t == 0) {
          __hip_atomic_store(dbg + 0, e - b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(dbg + 1, ne - nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(dbg + 2, e1 - b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(dbg + 3, me - mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          for (int k = 0; k < 4; k++) __hip_atomic_store(dbg + 4 + k, static_cast<int32_t>(sh->phase[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(dbg + 28, static_cast<int32_t>(sh->phase[4]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(dbg + 29, static_cast<int32_t>(sh->phase[5]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(dbg + 30, static_cast<int32_t>(sh->phase[6]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(dbg + 31, fresh ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
#endif
      LdsSync();
      if (n_dst >= 0) {
        float lec = s_x[n_dst - b] + n_a;  // the parenthesis of :309-311 was evaluated when the link was created
        if (!(lec > lb)) {
          if (lec < 0.0f) lec = 0.0f;
          __hip_atomic_fetch_min(&s_acc1[n_src - b], Enc(lec), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      for (int l = nb + NT + t; l < ne; l += NT) {
        const int dst = u.link_dst[l];
        if (dst < 0) continue;
        float lec = s_x[dst - b] + u.link_k[l];
        if (!(lec > lb)) {
          if (lec < 0.0f) lec = 0.0f;
          __hip_atomic_fetch_min(&s_acc1[u.link_src[l] - b], Enc(lec), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      LdsSync();
      bool changed = false;
      bool wm_nan = false;
      for (int i = t; i < e - b; i += NT) {
        const uint32_t a0 = s_acc0[i], a1 = s_acc1[i];
        const float v = Dec(a1 < a0 ? a1 : a0);
        changed |= !(v == s_x[i]);
        s_x[i] = v;
        s_acc1[i] = kEncInf;
        wm_nan |= v != v;
#ifdef KH_SERVE_MARKERS
        if (round_no == 5001 && changed && g_wave_mark != nullptr) {
          int32_t *r = g_wave_mark + blockIdx.x * 64 + 48;
          __hip_atomic_store(r + 0, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(r + 1, __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(r + 2, static_cast<int32_t>(a0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(r + 3, static_cast<int32_t>(a1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#endif
      }
      WM(28, (round_no & 0xfffff) | WM_ANY(wm_nan));
      if (u.phase_cycles != nullptr && t == 0) sh->phase[14] += 1;
      if (!BlockAny(changed, sh)) break;
      // The frame's epsilon links form a DAG no deeper than its tokens: a round beyond that means the links are not what
      // the closure wrote (round 6: one serving stream in ~3 % of the stress harness's runs spun here for ever and took the
      // whole grid's time-out with it).  The utterance fails (status 10) instead of the service.
      if (round_no > e - b + 8) { if (t == 0) sh->status = 10; break; }
    }
    WM(29, 0);
    // excise :315
    if (n_dst >= 0 && s_x[n_dst - b] + n_a > lb) {
      u.link_dst[nb + t] = -1;
      flags |= 2;
    }
    for (int l = nb + NT + t; l < ne; l += NT) {
      const int dst = u.link_dst[l];
      if (dst >= 0 && s_x[dst - b] + u.link_k[l] > lb) {
        u.link_dst[l] = -1;
        flags |= 2;
      }
    }
  } else {
    LdsSync();
  }
  // ---- write back; :334 counts the tokens whose extra_cost moved by more than delta
#pragma unroll
  for (int k = 0; k < kTPL; k++) {
    const int i = t + k * NT;
    if (i >= e - b || old_st[k] < 0) continue;
    const float old = old_x[k], v = s_x[i];
    if (!(v == old)) StoreExtra(&u.tok_extra[b + i], v);
    if (fabsf(v - old) > delta) flags |= 1;
  }
  const int all = BlockOr(flags, sh);
  *extra_costs_changed = (all & 1) != 0;
  *links_pruned = (all & 2) != 0;
}

// The same visit for a frame of up to 2 * kLdsSlots token slots (a frame that has not been compacted
// yet keeps the slots of its pruned tokens): only the extra_costs being computed live in LDS,
// ONE array relaxed in place — extra[t] starts as the minimum over t's emitting links and is
// lowered by atomic minima over the epsilon links until a sweep changes nothing.  The links of a
// frame form a DAG and min-plus relaxation is monotone, so any schedule reaches the same unique
// fixed point as the Jacobi rounds above, with the same float operations; cost[src], cost[dst] and
// extra[dst] are gathered from L2.  No global atomics (PruneForwardLinks keeps its epsilon
// accumulator in HBM: a DRAM read-modify-write per link and round).
__device__ void PruneFrameLdsBig(const Utt &u, const Params &p, int b, int e, int mb, int me, int nb, int ne, int b1, int e1,
                                 bool prune_toks_f1, bool fresh, float delta, bool *extra_costs_changed, bool *links_pruned, Blk &sh) {
  const float inf = INFINITY, lb = p.lattice_beam;
  const int t = KH_TIDX;
  // 2 * kLdsSlots words in two pieces (the value and the key array of the emitting pass's table): Enc(extra_cost) of f's tokens
  auto x_lo = LdsVals(sh);
  auto x_hi = LdsKeys(sh);
  auto x = [&](int i) -> __attribute__((address_space(3))) uint32_t * { return i < kLdsSlots ? &x_lo[i] : &x_hi[i - kLdsSlots]; };
  if (u.phase_cycles != nullptr && t == 0) { sh->phase[10] += 1; sh->phase[11] += e - b; sh->phase[14] += 1; }
  for (int i = t; i < e - b; i += NT) *x(i) = kEncInf;
  if (prune_toks_f1)   // PruneTokensForFrame(f + 1) :450-469: its extra_costs are final
    for (int i = b1 + t; i < e1; i += NT)
      if (LoadExtra(&u.tok_extra[i]) == inf && u.tok_state[i] >= 0) u.tok_state[i] = -1;
  LdsSync();
  // ---- emitting links (to frame f + 1, whose extra_costs are final): :309-323
  int flags = 0;
  constexpr int kBU = 4;   // a lane owns 4 consecutive links: one 16-byte access per array
  for (int l0 = mb + t * kBU; l0 < me; l0 += NT * kBU) {
    int dst[kBU], src[kBU];
    float kk[kBU];
    const bool full = l0 + kBU <= me;
    if (full) {
#ifndef KH_NO_NT
      const KhInt4 d4 = Load4I_NT(u.link_dst, l0), s4 = Load4I_NT(u.link_src, l0);
      const KhFloat4 k4 = Load4F_NT(u.link_k, l0);
#else
      const KhInt4 d4 = Load4I(u.link_dst, l0), s4 = Load4I(u.link_src, l0);
      const KhFloat4 k4 = Load4F(u.link_k, l0);
#endif
      dst[0] = d4.x; dst[1] = d4.y; dst[2] = d4.z; dst[3] = d4.w;
      src[0] = s4.x; src[1] = s4.y; src[2] = s4.z; src[3] = s4.w;
      kk[0] = k4.x; kk[1] = k4.y; kk[2] = k4.z; kk[3] = k4.w;
    } else {
#pragma unroll
      for (int k = 0; k < kBU; k++) {
        const int l = min(l0 + k, me - 1);
        dst[k] = u.link_dst[l]; src[k] = u.link_src[l]; kk[k] = u.link_k[l];
      }
    }
    float nx[kBU];
#pragma unroll
    for (int k = 0; k < kBU; k++) {
      nx[k] = 0.0f;
      if (dst[k] >= 0) {
        nx[k] = LoadExtra(&u.tok_extra[dst[k]]);
        // first visit: link_k still holds the candidate's tot_cost (:309-311's parenthesis = tot_cost - cost[dst])
        if (fresh) kk[k] = kk[k] - Dec(LoadCostEnc(&u.tok_cost[dst[k]]));
      }
    }
    bool killed = false;
#pragma unroll
    for (int k = 0; k < kBU; k++) {
      if (l0 + k >= me || dst[k] < 0) continue;
      float lec = nx[k] + kk[k];
      if (lec > lb) {
        dst[k] = -1;
        killed = true;
        if (!full) u.link_dst[l0 + k] = -1;
        flags |= 2;
      } else {
        if (fresh && !full) u.link_k[l0 + k] = kk[k];
        if (lec < 0.0f) lec = 0.0f;
        __hip_atomic_fetch_min(x(src[k] - b), Enc(lec), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (full) {
      if (killed) { KhInt4 o; o.x = dst[0]; o.y = dst[1]; o.z = dst[2]; o.w = dst[3]; Store4I(u.link_dst, l0, o); }
      // (the k of a link that died in this visit or earlier is never read again: any value may be stored)
      if (fresh) { KhFloat4 o; o.x = kk[0]; o.y = kk[1]; o.z = kk[2]; o.w = kk[3]; Store4F(u.link_k, l0, o); }
    }
  }
  // ---- epsilon links (inside the frame): relax in place to the fixed point
  if (ne > nb) {
    for (int round_no = 0;; round_no++) {
      WM(32, round_no);
      LdsSync();
      bool changed = false;
      for (int l = nb + t; l < ne; l += NT) {
        const int dst = u.link_dst[l];
        if (dst < 0) continue;
        float lec = Dec(*x(dst - b)) + u.link_k[l];  // the parenthesis of :309-311 was evaluated when the link was created
        if (!(lec > lb)) {
          if (lec < 0.0f) lec = 0.0f;
          const uint32_t v = Enc(lec);
          const uint32_t old = __hip_atomic_fetch_min(x(u.link_src[l] - b), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          changed |= v < old;
        }
      }
      if (u.phase_cycles != nullptr && t == 0) sh->phase[14] += 1;
      if (!BlockAny(changed, sh)) break;
    }
    // excise :315
    for (int l = nb + t; l < ne; l += NT) {
      const int dst = u.link_dst[l];
      if (dst >= 0 && Dec(*x(dst - b)) + u.link_k[l] > lb) {
        u.link_dst[l] = -1;
        flags |= 2;
      }
    }
  } else {
    LdsSync();
  }
  // ---- write back; :334 counts the tokens whose extra_cost moved by more than delta
  for (int i = t; i < e - b; i += NT) {
    if (u.tok_state[b + i] < 0) continue;
    const float old = LoadExtra(&u.tok_extra[b + i]), v = Dec(*x(i));
    if (!(v == old)) StoreExtra(&u.tok_extra[b + i], v);
    if (fabsf(v - old) > delta) flags |= 1;
  }
  const int all = BlockOr(flags, sh);
  *extra_costs_changed = (all & 1) != 0;
  *links_pruned = (all & 2) != 0;
}

// PruneTokensForFrame :450-469
__device__ void PruneTokensForFrame(const Utt &u, int b, int e) {
  for (int i = b + KH_TIDX; i < e; i += NT)
    if (u.tok_state[i] >= 0 && LoadExtra(&u.tok_extra[i]) == INFINITY) u.tok_state[i] = -1;
  KhSync();
}

// One backward-pruning visit of frame f (PruneForwardLinks(f) + PruneTokensForFrame(f + 1) when prune_toks_f1),
// dispatched by the frame's size to the LDS routines above.
__device__ __forceinline__ void PruneVisit(const Utt &u, const Params &p, int b, int e, int mb, int me, int nb, int ne, int b1, int e1,
                                           bool prune_toks_f1, bool fresh, float delta, bool *ec, bool *lp, Blk &sh) {
  WM(e - b <= kPruneLdsTok && e1 - b1 <= kPruneLdsTok ? 21 : (e - b <= 2 * kLdsSlots ? 22 : 23), e - b);
  if (e - b <= kPruneLdsTok && e1 - b1 <= kPruneLdsTok)
    PruneFrameLds(u, p, b, e, mb, me, nb, ne, b1, e1, prune_toks_f1, fresh, delta, ec, lp, sh);
  else if (e - b <= 2 * kLdsSlots)
    PruneFrameLdsBig(u, p, b, e, mb, me, nb, ne, b1, e1, prune_toks_f1, fresh, delta, ec, lp, sh);
  else
    PruneForwardLinks(u, p, b, e, mb, me, nb, ne, delta, false, false, 0.f, prune_toks_f1 ? b1 : 0, prune_toks_f1 ? e1 : 0, fresh, ec, lp, sh);
}

// PruneActiveTokens :476-503; cur = NumFramesDecoded().
template <class P>
__device__ __forceinline__ bool LoadFlag(P p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
template <class P>
__device__ __forceinline__ void StoreFlag(P p, uint8_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// final_pass: the backward loop of FinalizeDecoding :581-586 instead - every frame below `cur`, PruneForwardLinks(f, delta = 0)
// + PruneTokensForFrame(f + 1), whatever the flags say.
__device__ void PruneActiveTokens(const Utt &u, const Params &p, int cur, float delta, Blk &sh, bool final_pass = false) {
  // The bounds of frame `cur` were stored by thread 0 at the end of the previous frame's iteration, and when that frame's
  // closure ran in LDS (ClearHash returns at once) NO barrier lies between that store and this function's first visit,
  // which reads them (frame_b / frame_e[f + 1]) with one load per lane: a wave ahead of wave 0 read the words a previous
  // utterance had left there, took another routine or indexed LDS and memory with them.  That was round 6's serving
  // defect (one stream in ~3 % of the stress harness's runs spinning in PruneFrameLds, or a memory-access fault; always
  // the first visit; 77 runs of a debug build that happened to put a barrier here were clean, 4 of the last 20 without it
  // were not).
  KhSync();
  if (u.phase_cycles != nullptr && KH_TIDX == 0 && !final_pass) sh->phase[12] += 1;
  // every frame below conv_upto has been visited (a new frame has must_prune_forward_links set, and the
  // loop below cannot stop above it)
  const int conv_upto = Uni(sh->conv_upto);
  for (int f = cur - 1; f >= 0; f--) {
    // the flags and the frame's bounds in one round trip (the bounds are needed by most visits)
    const int ml_i = static_cast<int>(LoadFlag(&u.must_links[f]));
    const int mt_i = (f + 1 < cur) ? static_cast<int>(LoadFlag(&u.must_toks[f + 1])) : 0;
    const int vb = u.frame_b[f], ve = u.frame_e[f], vmb = u.femit_b[f], vme = u.femit_e[f], vnb = u.feps_b[f], vne = u.feps_e[f],
              vb1 = u.frame_b[f + 1], ve1 = u.frame_e[f + 1];
    const bool ml = final_pass || Uni(ml_i) != 0;
    const bool mt = final_pass || Uni(mt_i) != 0;
    // Flags of older frames can only be raised by the frame above them in this
    // pass (all frames visited by earlier passes were cleared), so once a frame
    // has nothing to do the reference's remaining iterations are no-ops.
    if (!ml && !mt) break;
    long long t0 = 0;
    if (u.phase_cycles != nullptr && KH_TIDX == 0) t0 = static_cast<long long>(__builtin_amdgcn_s_memtime());
    KM(20, f);
    WM(20, f);
    if (ml) {
      bool ec, lp;
      const bool fresh = f >= conv_upto;  // the frame's first visit: its emitting links still carry tot_cost in link_k
      const int b = Uni(vb), e = Uni(ve), mb = Uni(vmb), me = Uni(vme), nb = Uni(vnb), ne = Uni(vne), b1 = Uni(vb1), e1 = Uni(ve1);
      PruneVisit(u, p, b, e, mb, me, nb, ne, b1, e1, mt, fresh, delta, &ec, &lp, sh);
      WM(50, f);
      if (KH_TIDX == 0) {
        if (ec && f > 0) StoreFlag(&u.must_links[f - 1], 1);
        if (lp) StoreFlag(&u.must_toks[f], 1);
        StoreFlag(&u.must_links[f], 0);
        if (mt) StoreFlag(&u.must_toks[f + 1], 0);
      }
    } else {  // mt
      PruneTokensForFrame(u, Uni(vb1), Uni(ve1));
      if (KH_TIDX == 0) StoreFlag(&u.must_toks[f + 1], 0);
    }
    KhSync();
    if (u.phase_cycles != nullptr && KH_TIDX == 0) {
      const int thick = (Uni(u.frame_e[f]) - Uni(u.frame_b[f]) > NT) ? 1 : 0;
      sh->phase[16 + thick] += static_cast<long long>(__builtin_amdgcn_s_memtime()) - t0;
      sh->phase[18 + thick] += 1;
    }
  }
  if (KH_TIDX == 0) sh->conv_upto = cur;
  KhSync();
}

// ---------------------------------------------------------------- FinalizeDecoding, lazy schedule
// The backward loop of FinalizeDecoding :581-586 when it is followed by the export and nothing else (offline
// decoding): every frame is visited once, and all the visit has to produce is the list of what SURVIVES — about 17
// of a frame's ~5000 tokens and 27 of its ~8000 link slots.  Nothing is written back to the arenas (no excised
// marks, no converted link_k, no extra_costs, no pruned-token marks): a visit reads the frame's link slots once,
// keeps the frame's extra_costs in LDS (dense, up to kFinalDense tokens), hands the survivors' extra_costs to the
// next visit in a small LDS map (a link whose destination is not in it is dead: that is nearly every link, and it
// costs no memory access), and appends the survivors to the slot's survivor lists.  Same float operations and the
// same unique fixed point as PruneForwardLinks (rule P).  Frames beyond kFinalDense tokens, or with more survivors
// than the map holds, go through the general routines and global memory.
constexpr int kFinalDense = kLdsSlots + kLdsSlots / 2;   // 12288 extra_costs: the table's value array + half of its key array
constexpr int kMapSlots = 2048;                          // the other half: 2048 keys + 2048 values
constexpr int kMapMax = 1400;
static_assert(2 * kMapSlots == EU * NT * 2 && kLdsSlots / 2 == (1 << 11) + EU * NT, "the map lies over ex_ab + ex_tok, the dense array's upper part over hist + ex_off");

__device__ __forceinline__ void SurvAddTok(const Utt &u, Blk &sh, int i, int f) {
  const int pos = __hip_atomic_fetch_add(&sh->surv_nt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (pos < u.surv_tok_cap) { u.surv_tok[2 * pos] = i; u.surv_tok[2 * pos + 1] = f; }
  else sh->status = 7;
}
__device__ __forceinline__ void SurvAddLink(const Utt &u, Blk &sh, int l, int f) {
  const int pos = __hip_atomic_fetch_add(&sh->surv_nl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (pos < u.surv_link_cap) { u.surv_link[2 * pos] = l; u.surv_link[2 * pos + 1] = f; }
  else sh->status = 7;
}

__device__ void FinalBackward(const Utt &u_in, const Params &p, int last, int fb, int fe, Blk &sh) {
  // the pass's own copies of the pointers it uses (see ProcessEmitting: kept in scalar registers for its duration)
  Utt u = u_in;
  KH_LAUNDER_FB(u.link_dst.p); KH_LAUNDER_FB(u.link_src.p); KH_LAUNDER_FB(u.link_k.p); KH_LAUNDER_FB(u.tok_extra.p); KH_LAUNDER_FB(u.tok_state.p);
  KH_LAUNDER_FB(u.tok_cost.p); KH_LAUNDER_FB(u.surv_tok.p); KH_LAUNDER_FB(u.surv_link.p);
  KH_LAUNDER_FB(u.feps_b.p); KH_LAUNDER_FB(u.feps_e.p); KH_LAUNDER_FB(u.femit_b.p); KH_LAUNDER_FB(u.femit_e.p); KH_LAUNDER_FB(u.frame_b.p); KH_LAUNDER_FB(u.frame_e.p);
  const float inf = INFINITY, lb = p.lattice_beam;
  const int t = KH_TIDX;
  auto x_lo = LdsVals(sh);
  auto x_hi = LdsKeys(sh);
  auto x = [&](int i) -> __attribute__((address_space(3))) uint32_t * { return i < kLdsSlots ? &x_lo[i] : &x_hi[i - kLdsSlots]; };
  auto mk = reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(&sh->ex_ab[0]);   // key: token index within its frame + 1; 0 = empty
  auto mv = reinterpret_cast<__attribute__((address_space(3))) float *>(&sh->ex_tok[0]);
  auto map_slot = [](uint32_t i) { return (i * 0x9E3779B1u) >> 21; };
  const bool prof = u.phase_cycles != nullptr && t == 0;
  const int conv_upto = Uni(sh->conv_upto);
  // ---- the last frame (PruneForwardLinksFinal has run: extra_costs in memory, epsilon links excised)
  for (int i = fb + t; i < fe; i += NT)
    if (LoadExtra(&u.tok_extra[i]) != inf) SurvAddTok(u, sh, i, last);
  for (int l = Uni(u.feps_b[last]) + t; l < Uni(u.feps_e[last]); l += NT)
    if (u.link_dst[l] >= 0) SurvAddLink(u, sh, l, last);
  bool nx_global = true;   // extra_costs of frame f + 1: in tok_extra (true) or in the LDS map (false)
  int b1 = fb, e1 = fe;
  // bounds of the frame to visit, fetched one visit ahead
  int vb = 0, ve = 0, vmb = 0, vme = 0, vnb = 0, vne = 0;
  if (last > 0) { vb = u.frame_b[last - 1]; ve = u.frame_e[last - 1]; vmb = u.femit_b[last - 1]; vme = u.femit_e[last - 1]; vnb = u.feps_b[last - 1]; vne = u.feps_e[last - 1]; }
  KhSync();
  for (int f = last - 1; f >= 0; f--) {
    const int b = Uni(vb), e = Uni(ve), mb = Uni(vmb), me = Uni(vme), nb = Uni(vnb), ne = Uni(vne);
    if (f > 0) { vb = u.frame_b[f - 1]; ve = u.frame_e[f - 1]; vmb = u.femit_b[f - 1]; vme = u.femit_e[f - 1]; vnb = u.feps_b[f - 1]; vne = u.feps_e[f - 1]; }
    const int n = e - b;
    const bool fresh = f >= conv_upto;  // the frame's emitting links still carry tot_cost in link_k
    if (n > kFinalDense) {
      // ---- a frame too large for the dense array: the general routines on global memory (nx_global holds: the
      // visit before this one has seen this frame's size), then a sweep that lists what is left
      bool ec, lp;
      PruneVisit(u, p, b, e, mb, me, nb, ne, b1, e1, false, fresh, 0.0f, &ec, &lp, sh);
      KhSync();
      for (int i = b + t; i < e; i += NT)
        if (u.tok_state[i] >= 0 && LoadExtra(&u.tok_extra[i]) != inf) SurvAddTok(u, sh, i, f);
      for (int l = mb + t; l < me; l += NT)
        if (u.link_dst[l] >= 0) SurvAddLink(u, sh, l, f);
      for (int l = nb + t; l < ne; l += NT)
        if (u.link_dst[l] >= 0) SurvAddLink(u, sh, l, f);
      if (t == 0) sh->sched[2] += 1;
      nx_global = true;
      b1 = b; e1 = e;
      KhSync();
      continue;
    }
    if (t == 0) sh->sched[1] += 1;
    for (int i = t; i < n; i += NT) *x(i) = kEncInf;
    LdsSync();
    // ---- emitting links (to frame f + 1, whose extra_costs are final): :309-323
    constexpr int kBU = 4;   // a lane owns 4 consecutive link slots: one 16-byte access per array
    for (int l0 = mb + t * kBU; l0 < me; l0 += NT * kBU) {
      int dst[kBU], src[kBU];
      float kk[kBU];
      if (l0 + kBU <= me) {
#ifndef KH_NO_NT
        const KhInt4 d4 = Load4I_NT(u.link_dst, l0), s4 = Load4I_NT(u.link_src, l0);   // (the utterance's last pass over its links)
        const KhFloat4 k4 = Load4F_NT(u.link_k, l0);
#else
        const KhInt4 d4 = Load4I(u.link_dst, l0), s4 = Load4I(u.link_src, l0);
        const KhFloat4 k4 = Load4F(u.link_k, l0);
#endif
        dst[0] = d4.x; dst[1] = d4.y; dst[2] = d4.z; dst[3] = d4.w;
        src[0] = s4.x; src[1] = s4.y; src[2] = s4.z; src[3] = s4.w;
        kk[0] = k4.x; kk[1] = k4.y; kk[2] = k4.z; kk[3] = k4.w;
      } else {
#pragma unroll
        for (int k = 0; k < kBU; k++) {
          const int l = min(l0 + k, me - 1);
          dst[k] = l0 + k < me ? u.link_dst[l] : -1; src[k] = u.link_src[l]; kk[k] = u.link_k[l];
        }
      }
      float nx[kBU];
      if (nx_global) {
#pragma unroll
        for (int k = 0; k < kBU; k++) nx[k] = dst[k] >= 0 ? LoadExtra(&u.tok_extra[dst[k]]) : inf;
      } else {
#pragma unroll
        for (int k = 0; k < kBU; k++) {
          nx[k] = inf;
          if (dst[k] < 0) continue;
          const uint32_t key = static_cast<uint32_t>(dst[k] - b1) + 1u;
          uint32_t sl = map_slot(key);
          for (;;) {
            const uint32_t seen = mk[sl];
            if (seen == key) { nx[k] = mv[sl]; break; }
            if (seen == 0u) break;
            sl = (sl + 1) & (kMapSlots - 1);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < kBU; k++) {
        if (nx[k] == inf) continue;   // (also: an excised slot)
        // first visit: link_k still holds the candidate's tot_cost (:309-311's parenthesis = tot_cost - cost[dst])
        if (fresh) kk[k] = kk[k] - Dec(LoadCostEnc(&u.tok_cost[dst[k]]));
        float lec = nx[k] + kk[k];
        if (lec > lb) continue;
        if (lec < 0.0f) lec = 0.0f;
        __hip_atomic_fetch_min(x(src[k] - b), Enc(lec), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        SurvAddLink(u, sh, l0 + k, f);
      }
    }
    // ---- epsilon links (inside the frame): relax in place to the fixed point
    if (ne > nb && ne - nb <= NT) {
      // (nearly every frame: a few hundred slots) one link per lane, held in registers over the rounds; one barrier per
      // round - the vote's - which also orders the round's minima before the next round's reads, and does not wait
      // for the survivor lists' stores
      LdsSync();
      int dst = -1, src = b;
      float kk = 0.0f;
      if (t < ne - nb) { dst = u.link_dst[nb + t]; src = u.link_src[nb + t]; kk = u.link_k[nb + t]; }
      for (;;) {
        bool changed = false;
        if (dst >= 0) {
          float lec = Dec(*x(dst - b)) + kk;  // the parenthesis of :309-311 was evaluated when the link was created
          if (!(lec > lb)) {
            if (lec < 0.0f) lec = 0.0f;
            const uint32_t v = Enc(lec);
            const uint32_t old = __hip_atomic_fetch_min(x(src - b), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            changed = v < old;
          }
        }
        if (!BlockAnyLds(changed, sh)) break;
      }
      if (dst >= 0 && !(Dec(*x(dst - b)) + kk > lb)) SurvAddLink(u, sh, nb + t, f);
    } else if (ne > nb) {
      for (;;) {
        LdsSync();
        bool changed = false;
        for (int l = nb + t; l < ne; l += NT) {
          const int dst = u.link_dst[l];
          if (dst < 0) continue;
          float lec = Dec(*x(dst - b)) + u.link_k[l];  // the parenthesis of :309-311 was evaluated when the link was created
          if (!(lec > lb)) {
            if (lec < 0.0f) lec = 0.0f;
            const uint32_t v = Enc(lec);
            const uint32_t old = __hip_atomic_fetch_min(x(u.link_src[l] - b), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            changed |= v < old;
          }
        }
        if (!BlockAny(changed, sh)) break;
      }
      for (int l = nb + t; l < ne; l += NT) {
        const int dst = u.link_dst[l];
        if (dst >= 0 && !(Dec(*x(dst - b)) + u.link_k[l] > lb)) SurvAddLink(u, sh, l, f);
      }
    } else {
      LdsSync();
    }
    // ---- the frame's survivors: listed, and handed to the next visit.  (Every reader of the map of frame f + 1 is
    // behind a barrier by now.)
    const int n_next = f > 0 ? Uni(ve) - Uni(vb) : 0;
    bool to_global = n_next > kFinalDense;
    for (int i = t; i < kMapSlots; i += NT) mk[i] = 0u;
    if (t == 0) { sh->map_n = 0; sh->flag = 0; }
    LdsSync();
    if (!to_global) {
      for (int i = t; i < n; i += NT) {
        const uint32_t xe = *x(i);
        if (xe == kEncInf) continue;
        SurvAddTok(u, sh, b + i, f);
        if (__hip_atomic_fetch_add(&sh->map_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= kMapMax) { sh->flag = 1; continue; }
        const uint32_t key = static_cast<uint32_t>(i) + 1u;
        uint32_t sl = map_slot(key);
        for (;;) {
          uint32_t seen = 0u;
          __hip_atomic_compare_exchange_strong(&mk[sl], &seen, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (seen == 0u) break;
          sl = (sl + 1) & (kMapSlots - 1);
        }
        mv[sl] = Dec(xe);
      }
      LdsSync();
      to_global = Uni(sh->flag) != 0;
      if (to_global) {   // too many survivors for the map: the extra_costs go through memory (the survivors are listed already)
        for (int i = t; i < n; i += NT) StoreExtra(&u.tok_extra[b + i], Dec(*x(i)));
      }
    } else {
      for (int i = t; i < n; i += NT) {
        const uint32_t xe = *x(i);
        if (xe != kEncInf) SurvAddTok(u, sh, b + i, f);
        StoreExtra(&u.tok_extra[b + i], Dec(xe));
      }
    }
    if (t == 0 && to_global) sh->sched[3] += 1;
    nx_global = to_global;
    b1 = b; e1 = e;
    KhSync();
  }
  if (prof) { sh->phase[40] += sh->sched[1]; sh->phase[41] += sh->sched[2]; sh->phase[42] += sh->sched[3]; sh->phase[43] += sh->surv_nt; sh->phase[44] += sh->surv_nl; }
}

// In-place sliding compaction of the window [w_lo, cur]: survivors of the token
// arena tail and of the link arena tail move down, token indices stored in links
// are rewritten through tmp_remap (frame w_lo - 1 keeps its place but its emitting
// links point into the window, so their dst fields are rewritten too).  A chunk's
// slots are all read before the barrier of its scan and written at or below their
// old position after it, so one barrier per chunk orders the slide.
__device__ bool Compact(const Utt &u, int w_lo, int cur, bool keep_ac, Blk &sh) {
  if (w_lo < 0) w_lo = 0;
  const int win_b = Uni(u.frame_b[w_lo]);
  const int old_tok_end = Uni(sh->tok_end);
  const int old_link_b = Uni(u.feps_b[w_lo]);
  if (old_tok_end - win_b > u.window_cap) {
    if (KH_TIDX == 0) sh->status = 4;
    KhSync();
    return false;
  }
  KhSync();  // every thread has read the old ends
  long long tc = 0;
  const bool prof = u.phase_cycles != nullptr && KH_TIDX == 0;
  if (prof) tc = static_cast<long long>(__builtin_amdgcn_s_memtime());
#define KH_COMPACT_STAMP(k) do { if (prof) { const long long now_ = static_cast<long long>(__builtin_amdgcn_s_memtime()); sh->phase[k] += now_ - tc; tc = now_; } } while (0)
  // Both slides run over the window's whole RANGE, not block by block: a frame that is one or
  // two prune intervals old holds a few hundred slots, and a barrier per block was most of the
  // barriers of a call.  The blocks are contiguous in arena order, so a block's new bound is the
  // number of survivors before its old bound: the lane that owns that slot in the group's scan
  // writes it (bounds staged in LDS, kFB frames at a time so that a full compaction fits).
  auto bnd = sh->ex_off;   // old bounds (ascending)
  auto nbn = sh->ex_ab;    // new bounds
  constexpr int kFB = (EU * NT) / 2;
  // (a) tokens
  int tend = win_b;  // running end of the compacted tokens (uniform)
  {
    int base = win_b;
    for (int f0 = w_lo; f0 <= cur; f0 += kFB) {
      const int f1 = min(cur, f0 + kFB - 1), nb = f1 - f0 + 1;
      for (int j = KH_TIDX; j < nb; j += NT) bnd[j] = u.frame_b[f0 + j];
      const int chunk_e = f1 == cur ? old_tok_end : Uni(u.frame_b[f1 + 1]);
      KhSync();
      int bj = 0;
      while (base < chunk_e) {
        const int kk = min(KC, (base - tend) / NT);
        int off[KC], total, span;
        if (kk >= 2) {
          // only the liveness is read before the scan, the survivors' fields after it: safe when
          // none of the group's destinations [tend, tend + survivors) reaches into the group
          // itself, which a group of at most (gap / NT) chunks guarantees
          span = min(kk * NT, chunk_e - base);
          int st[KC], alive[KC];
#pragma unroll
          for (int k = 0; k < KC; k++) {
            const int i = base + k * NT + KH_TIDX;
            st[k] = (k < kk && i < chunk_e) ? u.tok_state[i] : -1;
          }
#pragma unroll
          for (int k = 0; k < KC; k++) alive[k] = st[k] >= 0 ? 1 : 0;
          BlockExScanK<KC, true>(alive, off, &total, sh);
#pragma unroll
          for (int k = 0; k < KC; k++) {
            const int i = base + k * NT + KH_TIDX;
            if (k >= kk || i >= chunk_e) continue;
            int ni = -1;
            if (alive[k]) {
              ni = tend + off[k];
              const uint32_t co = LoadCostEnc(&u.tok_cost[i]);
              const float ex = LoadExtra(&u.tok_extra[i]);
              u.tok_state[ni] = st[k]; u.tok_cost[ni] = co; u.tok_extra[ni] = ex;
            }
            u.tmp_remap[i - win_b] = ni;
          }
        } else {
          span = min(NT, chunk_e - base);
          const int i = base + KH_TIDX;
          int st = -1;
          uint32_t co = kEncInf;
          float ex = 0.f;
          if (i < chunk_e) { st = u.tok_state[i]; co = LoadCostEnc(&u.tok_cost[i]); ex = LoadExtra(&u.tok_extra[i]); }
          const int alive = st >= 0 ? 1 : 0;
          off[0] = BlockExScan(alive, &total, sh);
#pragma unroll
          for (int k = 1; k < KC; k++) off[k] = 0;
          if (i < chunk_e) {
            int ni = -1;
            if (alive) {
              ni = tend + off[0];
              u.tok_state[ni] = st; u.tok_cost[ni] = co; u.tok_extra[ni] = ex;
            }
            u.tmp_remap[i - win_b] = ni;
          }
        }
        while (bj < nb) {   // the bounds inside this group
          const int q = Uni(bnd[bj]) - base;
          if (q >= span) break;
          if (KH_TIDX == (q & (NT - 1))) {
            int o = off[0];
#pragma unroll
            for (int k = 1; k < KC; k++) o = (q / NT) == k ? off[k] : o;
            nbn[bj] = tend + o;
          }
          bj++;
        }
        tend += total;
        base += span;
      }
      for (; bj < nb; bj++)
        if (KH_TIDX == 0) nbn[bj] = tend;   // empty frames at the end of the batch
      KhSync();
      for (int j = KH_TIDX; j < nb; j += NT) {
        u.frame_b[f0 + j] = nbn[j];
        u.frame_e[f0 + j] = j + 1 < nb ? nbn[j + 1] : tend;
      }
      KhSync();  // (the staging arrays are reused by the next batch)
    }
  }
  KhSync();
  KH_COMPACT_STAMP(24);
  // arena invariant: free slots hold +inf
  for (int i = tend + KH_TIDX; i < old_tok_end; i += NT) u.tok_cost[i] = kEncInf;
  // (b) emitting links of frame w_lo - 1 point into the window: rewrite in place
  if (w_lo > 0) {
    for (int l = Uni(u.femit_b[w_lo - 1]) + KH_TIDX; l < Uni(u.femit_e[w_lo - 1]); l += NT) {
      const int dst = u.link_dst[l];
      if (dst >= win_b) u.link_dst[l] = u.tmp_remap[dst - win_b];
    }
  }
  KH_COMPACT_STAMP(25);
  // (c) links: one slide over eps(w_lo) emit(w_lo) eps(w_lo + 1) ... eps(cur)
  int lend = old_link_b;  // running end of the compacted links (uniform)
  {
    const int range_e = Uni(sh->link_end);
    int base = old_link_b;
    for (int f0 = w_lo; f0 <= cur; f0 += kFB) {
      const int f1 = min(cur, f0 + kFB - 1), nb = 2 * (f1 - f0 + 1);
      for (int j = KH_TIDX; j < nb; j += NT) {
        const int f = f0 + (j >> 1);
        // (the emitting block of `cur` is not created yet: empty, at the end of the range)
        bnd[j] = (j & 1) ? (f == cur ? range_e : u.femit_b[f]) : u.feps_b[f];
      }
      const int chunk_e = f1 == cur ? range_e : Uni(u.feps_b[f1 + 1]);
      KhSync();
      int bj = 0;
      while (base < chunk_e) {
        const int kk = min(KC, (base - lend) / NT);  // see the tokens above
        int off[KC], total, span;
        if (kk >= 2) {
          span = min(kk * NT, chunk_e - base);
          int dst[KC], alive[KC];
#pragma unroll
          for (int k = 0; k < KC; k++) {
            const int l = base + k * NT + KH_TIDX;
            dst[k] = (k < kk && l < chunk_e) ? u.link_dst[l] : -1;
          }
#pragma unroll
          for (int k = 0; k < KC; k++) alive[k] = dst[k] >= 0 ? 1 : 0;
          BlockExScanK<KC, true>(alive, off, &total, sh);
          if (prof) { sh->phase[35] += 1; sh->phase[36] += span; sh->phase[37] += total; }
          // three steps over the group so that the loads of its chunks are in flight together:
          // fields, then the remapped token indices (they depend on the fields), then the stores
          int src[KC], arc[KC];
          float g[KC], a[KC];
#pragma unroll
          for (int k = 0; k < KC; k++) {
            if (!alive[k]) continue;
            const int l = base + k * NT + KH_TIDX;
            src[k] = u.link_src[l]; arc[k] = u.link_arc[l];
            g[k] = u.link_k[l]; a[k] = keep_ac ? u.link_a[l] : 0.0f;
          }
#pragma unroll
          for (int k = 0; k < KC; k++) {
            if (!alive[k]) continue;
            if (dst[k] >= win_b) dst[k] = u.tmp_remap[dst[k] - win_b];
            if (src[k] >= win_b) src[k] = u.tmp_remap[src[k] - win_b];
          }
#pragma unroll
          for (int k = 0; k < KC; k++) {
            if (!alive[k]) continue;
            const int d = lend + off[k];
            u.link_dst[d] = dst[k]; u.link_src[d] = src[k];
            u.link_arc[d] = arc[k]; u.link_k[d] = g[k];
            if (keep_ac) u.link_a[d] = a[k];
          }
        } else {
          span = min(NT, chunk_e - base);
          const int l = base + KH_TIDX;
          int dst = -1, src = 0, arc = 0;
          float g = 0.f, a = 0.f;
          if (l < chunk_e) {
            dst = u.link_dst[l];
            if (dst >= 0) { src = u.link_src[l]; arc = u.link_arc[l]; g = u.link_k[l]; a = keep_ac ? u.link_a[l] : 0.0f; }
          }
          const int alive = dst >= 0 ? 1 : 0;
          off[0] = BlockExScan(alive, &total, sh);
#pragma unroll
          for (int k = 1; k < KC; k++) off[k] = 0;
          if (prof) { sh->phase[34] += 1; sh->phase[36] += span; sh->phase[37] += total; }
          if (alive) {
            const int d = lend + off[0];
            u.link_dst[d] = dst >= win_b ? u.tmp_remap[dst - win_b] : dst;
            u.link_src[d] = src >= win_b ? u.tmp_remap[src - win_b] : src;
            u.link_arc[d] = arc; u.link_k[d] = g;
            if (keep_ac) u.link_a[d] = a;
          }
        }
        while (bj < nb) {   // the block bounds inside this group
          const int q = Uni(bnd[bj]) - base;
          if (q >= span) break;
          if (KH_TIDX == (q & (NT - 1))) {
            int o = off[0];
#pragma unroll
            for (int k = 1; k < KC; k++) o = (q / NT) == k ? off[k] : o;
            nbn[bj] = lend + o;
          }
          bj++;
        }
        lend += total;
        base += span;
      }
      for (; bj < nb; bj++)
        if (KH_TIDX == 0) nbn[bj] = lend;   // empty blocks at the end of the batch
      KhSync();
      for (int j = KH_TIDX; j < nb; j += NT) {
        const int f = f0 + (j >> 1);
        const int nbeg = nbn[j], nend = j + 1 < nb ? nbn[j + 1] : lend;
        if (j & 1) {
          if (f != cur) { u.femit_b[f] = nbeg; u.femit_e[f] = nend; }
        } else {
          u.feps_b[f] = nbeg; u.feps_e[f] = nend;
        }
      }
      KhSync();
    }
  }
  if (KH_TIDX == 0) {
    sh->tok_end = tend;
    sh->link_end = lend;
    sh->front_b = Uni(u.frame_b[cur]);
  }
  KhSync();
  KH_COMPACT_STAMP(26);
#undef KH_COMPACT_STAMP
  return true;
}

// Frontier of a decoding run (uniform over the workgroup): frames decoded so far and
// the token range of the newest frame.
struct Run {
  int t, fb, fe;
  long long cand;   // emitting candidates materialised so far (uniform: kept in scalar registers, not in LDS - an LDS
                    // accumulator updated inside ProcessEmitting cost the kernel 5 % through its register allocation)
};

// Compaction window: everything younger than 2 * max(prune_interval, 25) frames
// (frames leave it only once they are >= 25 frames behind the frontier, i.e.
// thinned to lattice density by the backward pruning).
#ifndef KH_COMPACT_EVERY
#define KH_COMPACT_EVERY 2   // the window is compacted at every KH_COMPACT_EVERY-th call of PruneActiveTokens
#endif
__device__ __forceinline__ int WindowFrames(const Params &p) { return (KH_COMPACT_EVERY + 1) * (p.prune_interval > 25 ? p.prune_interval : 25); }

// InitDecoding :55-72 on a slot whose arenas hold their invariants.
template <bool kExact = false>
__device__ bool DecodeInit(const Utt &u, const Params &p, Blk &sh, Run *run) {
  if (KH_TIDX == 0) {
    sh->tok_end = 0;
    sh->link_end = 0;
    sh->front_b = 0;
    sh->conv_upto = 0;
    sh->status = 0;
    sh->arcs_expanded = 0;
    sh->tokens_created = 0;
    sh->max_tokens_frame = 0;
    sh->gc_tok = 0;
    sh->gc_link = 0;
    sh->surv_nt = 0;
    sh->surv_nl = 0;
    for (int i = 0; i < 4; i++) sh->sched[i] = 0;
  }
  for (int f = KH_TIDX; f < u.T + 2; f += NT) {
    u.must_links[f] = 1;  // TokenList(): must_prune_forward_links(true), must_prune_tokens(true)
    u.must_toks[f] = 1;
  }
  KhSync();
  if (KH_TIDX == 0) {
    sh->eps_n = 0;
    sh->erel_n = 0;
    sh->hash_dirty = 1;
    const int idx = FindOrAdd<true>(u, p.start, p.start_has_eps != 0, &sh->tok_end, &sh->eps_n, u.tok_cap, 0);
    u.tok_cost[idx] = Enc(0.0f);
    u.frame_b[0] = idx;
    sh->wl_n[0] = sh->eps_n;  // 1 if the start state has epsilon arcs: the closure's first work list
    sh->wl_n[1] = 0;
    if (kExact) {   // the start token is the first insertion (:66) into a table of 1000 buckets (:37)
      UX(x_q)[0] = 0u;
      UX(x_c0e)[0] = Enc(0.0f);
      UX(x_bkt)[0] = -1 - p.unit_ilabel[p.start];   // the start state's own id
      sh->x_hsize = 1000u;
      sh->x_qbase = 1u;
      sh->x_ne_emit = sh->tok_end;
      sh->x_eps_emit = sh->eps_n;
    }
  }
  KhSync();
  bool ok = ProcessNonemitting(u, p, 0, p.beam, sh, false);
  if (kExact && ok) ok = OrderFrontier(u, p, 0, Uni(sh->tok_end), Uni(u.feps_b[0]), Uni(u.feps_e[0]), p.beam, sh);
  run->t = 0;
  run->cand = 0;
  run->fb = 0;  // token range of the frontier frame
  run->fe = Uni(sh->tok_end);
  if (KH_TIDX == 0) {
    u.frame_e[0] = run->fe;
    sh->tokens_created += run->fe - run->fb;
    if (run->fe > sh->tok_hw) sh->tok_hw = run->fe;
  }
  if (ok) ClearHash(u, run->fb, run->fe, sh);
  return ok;
}

#ifdef KH_SERVE_MARKERS
// (debug build) the links and tokens a frame has just created, checked where they are created: a violation is written to
// the stream's debug block (words 16..) and the workgroup STAYS here - the host's time-out dump then shows it.
__device__ void DbgCheckFrame(const Utt &u, const Params &p, Blk &sh, int t, int t_start, int pfb, int pfe, int fb, int fe) {
  if (g_wave_mark == nullptr) return;
  int32_t *dbg = g_wave_mark + blockIdx.x * 64 + 16;
  const int mb = Uni(u.femit_b[t]), me = Uni(u.femit_e[t]), nb = Uni(u.feps_b[t + 1]), ne = Uni(u.feps_e[t + 1]);
  int cat = -1, wl = 0, wd = 0, ws = 0, wk = 0;
  for (int l = mb + KH_TIDX; l < me; l += NT) {
    const int dst = u.link_dst[l], src = u.link_src[l];
    const float kv = u.link_k[l];
    if (dst < 0) continue;
    int c = -1;
    if (dst < fb || dst >= fe) c = 10; else if (src < pfb || src >= pfe) c = 11; else if (!(kv == kv) || !(fabsf(kv) < INFINITY)) c = 12;
    if (c >= 0) { cat = c; wl = l - mb; wd = dst; ws = src; wk = __float_as_int(kv); }
  }
  for (int l = nb + KH_TIDX; l < ne; l += NT) {
    const int dst = u.link_dst[l], src = u.link_src[l];
    const float kv = u.link_k[l];
    if (dst < 0) continue;
    int c = -1;
    if (dst < fb || dst >= fe) c = 20; else if (src < fb || src >= fe) c = 21; else if (!(kv >= -1.0e-3f) || !(kv < INFINITY)) c = 22;
    if (c >= 0) { cat = c; wl = l - nb; wd = dst; ws = src; wk = __float_as_int(kv); }
  }
  for (int i = fb + KH_TIDX; i < fe; i += NT) {
    const float cv = Dec(LoadCostEnc(&u.tok_cost[i]));
    const int st = u.tok_state[i];
    int c = -1;
    if (!(cv == cv) || !(fabsf(cv) < INFINITY)) c = 30; else if (st < 0 || st >= p.num_units) c = 31;
    if (c >= 0) { cat = c; wl = i - fb; wd = st; ws = 0; wk = __float_as_int(cv); }
  }
  if (cat >= 0) {
    __hip_atomic_store(dbg + 1, wl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dbg + 2, wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dbg + 3, ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dbg + 4, wk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dbg + 0, 7000 + cat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (BlockAny(cat >= 0, sh)) {
    if (KH_TIDX == 0) {
      const int vals[11] = {t, t_start, pfb, pfe, fb, fe, mb, me, nb, ne, static_cast<int>(Uni(sh->hash_dirty))};
      for (int k = 0; k < 11; k++) __hip_atomic_store(dbg + 5 + k, vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (;;) __builtin_amdgcn_s_sleep(127);
  }
}
#endif

// Decode :77-95 / AdvanceDecoding (lattice-faster-online-decoder.cc:747-769): frames
// [run->t, t_end).  u.ll is addressed by absolute frame.
template <bool kLazy, bool kExact = false>
__device__ bool DecodeFrames(const Utt &u, const Params &p, Blk &sh, Run *run, int t_end) {
  const int win_frames = WindowFrames(p);
  bool ok = true;
  int t = run->t, fb = run->fb, fe = run->fe;
  long long cand = run->cand;
  int last_gc = 0;  // (lazy schedule) frame of the last garbage collection
  for (; ok && t < t_end; t++) {
    KM(1, t);
    if (kLazy) {
      // Garbage collection on demand: the next frame may take up to tok_frame_cap tokens and link_frame_cap emitting +
      // link_frame_cap epsilon link slots.  PruneActiveTokens visits every frame since the last collection for the first
      // time (their must_prune flags are still set) and then walks back as far as extra_costs keep moving; the full
      // compaction slides the survivors down.  Not more often than every prune_interval frames: an arena that is full of
      // LIVE data reports its overflow instead (the utterance is decoded again with larger arenas).
      const bool low = u.tok_cap - Uni(sh->tok_end) < u.tok_frame_cap || u.link_cap - Uni(sh->link_end) < 2 * u.link_frame_cap;
      const bool span = p.lazy_span > 0 && t - Uni(sh->conv_upto) >= p.lazy_span;   // (conv_upto: the frame of the last collection)
      if ((span && t > 0) || (low && t > 0 && t - last_gc >= p.prune_interval)) {
        Stamp(u, sh, 15);
        PruneActiveTokens(u, p, t, p.lattice_beam * p.prune_scale, sh);
        Stamp(u, sh, 6);
        ok = Compact(u, 0, t, p.keep_ac != 0, sh);
        if (KH_TIDX == 0) { sh->gc_tok = sh->tok_end; sh->gc_link = sh->link_end; }
        KhSync();
        Stamp(u, sh, 7);
        if (!ok) break;
        last_gc = t;
        if (KH_TIDX == 0) sh->sched[0] += 1;
        fb = Uni(u.frame_b[t]);
        fe = Uni(u.frame_e[t]);
      }
    } else if (t % p.prune_interval == 0 && t > 0) {
      Stamp(u, sh, 15);
      KM(2, t);
      PruneActiveTokens(u, p, t, p.lattice_beam * p.prune_scale, sh);
      KM(3, t);
      Stamp(u, sh, 6);
      if (Uni(sh->status) != 0) { ok = false; break; }   // (status 10: a frame whose links are inconsistent)
      if ((t / p.prune_interval) % KH_COMPACT_EVERY == 0) ok = Compact(u, t - win_frames, t, p.keep_ac != 0, sh);
      // Frames older than the window keep the slots of what was pruned after they left it
      // (the backward pruning keeps thinning frames ~200 frames behind the frontier): once that
      // garbage has grown to a third of an arena, compact everything (rare: every ~700 frames
      // of a long utterance; a full sweep costs about a dozen frames of decoding).
      if (ok && (Uni(sh->tok_end) - Uni(sh->gc_tok) > u.tok_cap / 3 || Uni(sh->link_end) - Uni(sh->gc_link) > u.link_cap / 3)) {
        ok = Compact(u, 0, t, p.keep_ac != 0, sh);
        if (KH_TIDX == 0) { sh->gc_tok = sh->tok_end; sh->gc_link = sh->link_end; }
        KhSync();
      }
      Stamp(u, sh, 7);
      KM(4, t);
      if (!ok) break;
      fb = Uni(u.frame_b[t]);
      fe = Uni(u.frame_e[t]);
    }
    float next_cutoff;
    int n_cand = 0;
#ifdef KH_SERVE_MARKERS
    const int dbg_pfb = fb, dbg_pfe = fe;
#endif
    ok = kExact ? ProcessEmittingExact(u, p, t, fb, fe, &next_cutoff, &n_cand, sh) : ProcessEmitting(u, p, t, fb, fe, &next_cutoff, &n_cand, sh);
    KM(5, t);
    if (!ok) break;
    cand += n_cand;
    ok = ProcessNonemitting(u, p, t + 1, next_cutoff, sh);
    KM(6, t);
    if (!ok) break;
    fb = Uni(sh->front_b);
    fe = Uni(sh->tok_end);
    if (kExact) {   // the new frame's list positions (the order the reference would walk it in)
      Stamp(u, sh, 4);
      ok = OrderFrontier(u, p, fb, fe, Uni(u.feps_b[t + 1]), Uni(u.feps_e[t + 1]), next_cutoff, sh);
      Stamp(u, sh, 45);
      XS(28);
      KM(7, t);
      if (!ok) break;
    }
    if (KH_TIDX == 0) {
      u.frame_b[t + 1] = fb;
      u.frame_e[t + 1] = fe;
      sh->tokens_created += fe - fb;
      if (fe > sh->tok_hw) sh->tok_hw = fe;
    }
    Stamp(u, sh, 15);
    ClearHash(u, fb, fe, sh);
    Stamp(u, sh, 5);
    XS(29);
#ifdef KH_SERVE_MARKERS
    KhSync();
    DbgCheckFrame(u, p, sh, t, run->t, dbg_pfb, dbg_pfe, fb, fe);
#endif
  }
  run->t = t;
  run->fb = fb;
  run->fe = fe;
  run->cand = cand;
  return ok;
}

// FinalizeDecoding :573-588 (ComputeFinalCosts :505-545 first) after run->t frames,
// then the counters of the utterance.  Leaves the surviving tokens/links in the
// slot's arenas ([0, sh->tok_end) / [0, sh->link_end)).
template <bool kLazy>
__device__ bool DecodeFinalize(const Utt &u, const Params &p, Blk &sh, const Run &run, bool ok, KhDecodeStats *st_out) {
  const float inf = INFINITY;
  const int fb = run.fb, fe = run.fe;
  KhDecodeStats st;
  st.num_frames = run.t;
  st.reached_final = 0;
  st.final_relative_cost = inf;
  st.final_best_cost = inf;
  st.num_tokens = 0;
  st.num_links = 0;
  if (ok) {
    const int last = run.t;
    float best_cost = inf, best_with_final = inf;
    for (int i = fb + KH_TIDX; i < fe; i += NT) {
      const float cost = Dec(LoadCostEnc(&u.tok_cost[i]));
      const float final_cost = __int_as_float(p.rec[u.tok_state[i]].w);
      best_cost = fminf(best_cost, cost);
      best_with_final = fminf(best_with_final, cost + final_cost);
    }
    best_cost = BlockMinF(best_cost, sh);
    best_with_final = BlockMinF(best_with_final, sh);
    const bool have_final = best_with_final != inf;  // final_costs_ non-empty
    st.reached_final = have_final ? 1 : 0;
    st.final_relative_cost = (best_cost == inf && best_with_final == inf) ? inf : best_with_final - best_cost;
    const float final_best_cost = have_final ? best_with_final : best_cost;
    st.final_best_cost = final_best_cost;
    bool b1, b2;
    PruneForwardLinks(u, p, fb, fe, 0, 0, Uni(u.feps_b[last]), Uni(u.feps_e[last]), 0.0f, true, have_final, final_best_cost,
                      0, 0, false, &b1, &b2, sh);
    // :581-586 every frame once
    KhSync();  // the last frame's extra_costs are in place
    if (kLazy) {
      // most frames are visited for the first time, at their full size, and only the survivor lists are produced
      // (ExportSurvivors copies them): the arenas keep their pre-pruning state
      FinalBackward(u, p, last, fb, fe, sh);
    } else {
      PruneActiveTokens(u, p, last, 0.0f, sh, true);
      if (KH_TIDX == 0) sh->conv_upto = last;
      PruneTokensForFrame(u, Uni(u.frame_b[0]), Uni(u.frame_e[0]));
      // final compaction of the window so the export below copies little
      ok = Compact(u, last - WindowFrames(p), last, p.keep_ac != 0, sh);
    }
    Stamp(u, sh, 8);
  }
  if (KH_TIDX == 0) sh->cand_mat = run.cand;
  KhSync();
  st.arcs_expanded = sh->arcs_expanded;
  st.tokens_created = sh->tokens_created;
  st.status = sh->status;
  st.max_tokens_frame = sh->max_tokens_frame;
  st.num_tokens = sh->tok_end;   // arena slots in use (pruned ones included)
  st.num_links = sh->link_end;
  *st_out = st;
  KhSync();
  return ok && Uni(sh->status) == 0;
}

// One utterance: InitDecoding, Decode, FinalizeDecoding.
template <bool kLazy, bool kExact>
__device__ bool DecodeOne(const Utt &u, const Params &p, Blk &sh, KhDecodeStats *st_out) {
  Run run;
  bool ok = DecodeInit<kExact>(u, p, sh, &run);
  if (ok) ok = DecodeFrames<kLazy, kExact>(u, p, sh, &run, u.T);
  return DecodeFinalize<kLazy>(u, p, sh, run, ok, st_out);
}

// Per-utterance inputs / outputs of the batch and the lattice pool the finished
// utterances are exported to (so that the slot's arenas can be reused).
struct UttIn {
  GP(const float) ll;
  int32_t T, pad;
};
struct UttOut {
  KhDecodeStats stats;
  long long tok_off, link_off;  // position in the pool
  int32_t n_tok, n_link;
  int32_t sched[4];             // Shared::sched of the utterance
  long long cand_mat;           // emitting candidates materialised (got a link slot)
};
struct Pool {
  GP(int32_t) t_frame; GP(int32_t) t_state;   // per exported token
  GP(int32_t) l_src; GP(int32_t) l_dst; GP(int32_t) l_il; GP(int32_t) l_ol;  // per exported link (utterance-relative)
  GP(float) l_g; GP(float) l_a;               // graph cost, acoustic cost - cost_offset[frame]
  long long tok_cap, link_cap;
  GP(unsigned long long) used;                // [0] tokens, [1] links, [2] utterance queue head
};

// Frame of token i: the f with frame_b[f] <= i < frame_e[f] (frames are contiguous
// and ordered; empty frames share their begin with the next one).
__device__ __forceinline__ int FrameOfToken(const Utt &u, int i, int T) {
  int lo = 0, hi = T;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (u.frame_b[mid] <= i) lo = mid; else hi = mid - 1;
  }
  while (lo < T && u.frame_e[lo] <= i) lo++;
  return lo;
}

// GetRawLattice :109-191 device half: survivors -> pool (frame, state) / (src, dst, labels, costs).
__device__ void ExportLattice(const Utt &u, const Params &p, const Pool &pool, UttOut *out, Blk &sh) {
  const int tok_end = Uni(sh->tok_end), link_end = Uni(sh->link_end), T = u.T;
  // pass A: alive tokens -> dense indices (tmp_remap)
  int n_tok = 0;
  for (int base = 0; base < tok_end; base += NT) {
    const int i = base + KH_TIDX;
    const int alive = (i < tok_end && u.tok_state[i] >= 0) ? 1 : 0;
    int total;
    const int off = BlockExScan(alive, &total, sh);
    if (alive) u.tmp_remap[i] = n_tok + off;   // (a live link's tokens are alive: dead slots are never looked up)
    n_tok += total;
  }
  // pass B: alive links -> dense positions, kept in link_src's dead... count only
  int n_link = 0;
  {
    int mine = 0;
    for (int l = KH_TIDX; l < link_end; l += NT) mine += u.link_dst[l] >= 0 ? 1 : 0;
    n_link = static_cast<int>(BlockSumLL(mine, sh));
  }
  // allocate in the pool
  if (KH_TIDX == 0) {
    const unsigned long long tb = __hip_atomic_fetch_add(&pool.used[0], static_cast<unsigned long long>(n_tok), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long lb = __hip_atomic_fetch_add(&pool.used[1], static_cast<unsigned long long>(n_link), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out->tok_off = static_cast<long long>(tb);
    out->link_off = static_cast<long long>(lb);
    out->n_tok = n_tok;
    out->n_link = n_link;
    const bool fits = tb + n_tok <= static_cast<unsigned long long>(pool.tok_cap) &&
                      lb + n_link <= static_cast<unsigned long long>(pool.link_cap);
    if (!fits) { sh->status = 6; out->stats.status = 6; }
    sh->wmin[0] = tb;
    sh->wmin[1] = lb;
    sh->flag = fits ? 1 : 0;
  }
  KhSync();
  const long long tb = static_cast<long long>(sh->wmin[0]), lbase = static_cast<long long>(sh->wmin[1]);
  const bool fits = sh->flag != 0;
  KhSync();
  if (!fits) return;
  // pass C: tokens
  for (int i = KH_TIDX; i < tok_end; i += NT) {
    const int st = u.tok_state[i];
    if (st < 0) continue;
    const int ni = u.tmp_remap[i];
    pool.t_frame[tb + ni] = FrameOfToken(u, i, T);
    pool.t_state[tb + ni] = -1 - p.unit_ilabel[st];   // the caller's state id (kept in the header's slot of the label table)
  }
  // pass D: links, in arena order
  int lrun = 0;
  for (int base = 0; base < link_end; base += NT) {
    const int l = base + KH_TIDX;
    const int dst = l < link_end ? u.link_dst[l] : -1;
    const int alive = dst >= 0 ? 1 : 0;
    int total;
    const int off = BlockExScan(alive, &total, sh);
    if (alive) {
      const long long d = lbase + lrun + off;
      const int src = u.link_src[l], arc = u.link_arc[l];
      const KhInt4 rec = arc >= 0 ? p.rec[arc] : p.n_arcs[-1 - arc];   // labels: from the arc (3 % of the links survive to here)
      const int il = arc >= 0 ? p.unit_ilabel[arc] : 0;
      float a = 0.0f;  // (an epsilon link has none)
      if (il != 0) {  // :168-174 the acoustic cost without the frame's cost_offset
        const int f = FrameOfToken(u, src, T);
        const float co = f < T ? u.cost_offset[f] : 0.0f;
        // ac_cost = cost_offset - loglike (:724-725): stored with the link, or evaluated again from the score matrix
        int pdf = rec.x;
        KH_BOUND(8, pdf, 0, u.ll_stride);
        a = p.keep_ac ? u.link_a[l] : (f < T ? co - u.ll[static_cast<size_t>(f) * u.ll_stride + pdf] : 0.0f);
        a -= co;
      }
      pool.l_src[d] = u.tmp_remap[src];
      pool.l_dst[d] = u.tmp_remap[dst];
      pool.l_il[d] = il;
      pool.l_ol[d] = (sh.x != nullptr && arc >= 0) ? UX(x_rec0)[arc].y : rec.y;   // (reference order: see ArcPdfKernel)
      pool.l_g[d] = __int_as_float(rec.z);   // the graph cost is the arc's weight
      pool.l_a[d] = a;
    }
    lrun += total;
  }
  KhSync();
}

// GetRawLattice :109-191 device half from the survivor lists of FinalBackward (lazy schedule): the arenas are not scanned.
__device__ void ExportSurvivors(const Utt &u, const Params &p, const Pool &pool, UttOut *out, Blk &sh) {
  const int n_tok = Uni(sh->surv_nt), n_link = Uni(sh->surv_nl), T = u.T;
  if (KH_TIDX == 0) {
    const unsigned long long tb = __hip_atomic_fetch_add(&pool.used[0], static_cast<unsigned long long>(n_tok), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long lb = __hip_atomic_fetch_add(&pool.used[1], static_cast<unsigned long long>(n_link), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out->tok_off = static_cast<long long>(tb);
    out->link_off = static_cast<long long>(lb);
    out->n_tok = n_tok;
    out->n_link = n_link;
    const bool fits = tb + n_tok <= static_cast<unsigned long long>(pool.tok_cap) &&
                      lb + n_link <= static_cast<unsigned long long>(pool.link_cap);
    if (!fits) { sh->status = 6; out->stats.status = 6; }
    sh->wmin[0] = tb;
    sh->wmin[1] = lb;
    sh->flag = fits ? 1 : 0;
  }
  KhSync();
  const long long tb = static_cast<long long>(sh->wmin[0]), lbase = static_cast<long long>(sh->wmin[1]);
  const bool fits = sh->flag != 0;
  KhSync();
  if (!fits) return;
  for (int j = KH_TIDX; j < n_tok; j += NT) {
    const int i = u.surv_tok[2 * j], f = u.surv_tok[2 * j + 1];
    u.tmp_remap[i] = j;
    pool.t_frame[tb + j] = f;
    pool.t_state[tb + j] = -1 - p.unit_ilabel[u.tok_state[i]];   // the caller's state id (kept in the header's slot of the label table)
  }
  KhSync();
  for (int j = KH_TIDX; j < n_link; j += NT) {
    const int l = u.surv_link[2 * j], f = u.surv_link[2 * j + 1];
    const long long d = lbase + j;
    const int src = u.link_src[l], dst = u.link_dst[l], arc = u.link_arc[l];
    const KhInt4 rec = arc >= 0 ? p.rec[arc] : p.n_arcs[-1 - arc];
    const int il = arc >= 0 ? p.unit_ilabel[arc] : 0;
    float a = 0.0f;  // (an epsilon link has none)
    if (il != 0) {  // :168-174 the acoustic cost without the frame's cost_offset
      const float co = f < T ? u.cost_offset[f] : 0.0f;
      int pdf = rec.x;
      KH_BOUND(8, pdf, 0, u.ll_stride);
      a = p.keep_ac ? u.link_a[l] : (f < T ? co - u.ll[static_cast<size_t>(f) * u.ll_stride + pdf] : 0.0f);
      a -= co;
    }
    pool.l_src[d] = u.tmp_remap[src];
    pool.l_dst[d] = u.tmp_remap[dst];
    pool.l_il[d] = il;
    pool.l_ol[d] = (sh.x != nullptr && arc >= 0) ? UX(x_rec0)[arc].y : rec.y;   // (reference order: see ArcPdfKernel)
    pool.l_g[d] = __int_as_float(rec.z);   // the graph cost is the arc's weight
    pool.l_a[d] = a;
  }
  KhSync();
}

// Persistent workgroups: each owns one slot (arena set) and pulls utterances from
// a queue (the host orders them longest-first) until it is empty.
#ifndef KH_WG_PER_CU
#define KH_WG_PER_CU 2   // two 1024-thread workgroups per CU (<= 64 VGPRs): more loads in flight
#endif
template <bool kLazy, bool kExact>
__global__ void __launch_bounds__(NT)
#if KH_WG_PER_CU > 1
__attribute__((amdgpu_waves_per_eu(NT / 256 * KH_WG_PER_CU, NT / 256 * KH_WG_PER_CU)))
#endif
DecodeKernel(const Utt *__restrict__ slots, const UttIn *__restrict__ in, UttOut *__restrict__ out,
             int n_utts, Pool pool, Params p, GP(long long) phase_cycles, GP(int32_t) done_list, const UttX *slotsx) {
  __shared__ Shared shm;
  extern __shared__ float dyn_ll_row[];
  Blk sh;
  sh.p = (LdsShared *)&shm;
  sh.ll_row = (__attribute__((address_space(3))) float *)dyn_ll_row;
  InitCalls(sh, kExact);
  sh.x = kExact ? (__attribute__((address_space(4))) const UttX *)(slotsx + blockIdx.x) : nullptr;
  Utt u = slots[blockIdx.x];
  Launder(u);
  Launder(p);
  u.phase_cycles = phase_cycles ? phase_cycles + NPH * blockIdx.x : (GP(long long))nullptr;
  if (KH_TIDX == 0) {
    for (int i = 0; i < NPH; i++) sh->phase[i] = 0;
    sh->t_last = static_cast<long long>(__builtin_amdgcn_s_memtime());
    sh->tok_hw = 0;
    for (int i = 0; i < 4; i++) sh->orbuf[i] = 0;
    sh->hash_dirty = 0;
#ifdef KH_BARRIER_CHECK
    for (int i = 0; i < 16; i++) __hip_atomic_store(&g_bar_cnt[blockIdx.x * 16 + i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  }
#ifdef KH_BARRIER_CHECK
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#endif
  KhSync();
  for (;;) {
    if (KH_TIDX == 0) sh->bcast_i[3] = static_cast<int>(__hip_atomic_fetch_add(&pool.used[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    KhSync();
    const int ui = Uni(sh->bcast_i[3]);
    KhSync();
    if (ui >= n_utts) break;
    u.ll = in[ui].ll;
    u.T = in[ui].T;
    // restore the arena invariants left dirty by the previous utterance of this slot
    const int hw = sh->tok_hw;
    for (int i = KH_TIDX; i < hw; i += NT) u.tok_cost[i] = kEncInf;
    for (uint32_t i = KH_TIDX; i <= u.hash_mask; i += NT) u.hash[i] = kEmpty;
    // (a capacity overflow aborts a frame with work-list flags still set: clear them too)
    for (int i = KH_TIDX; i < u.tok_frame_cap; i += NT) { u.tmp_acc1[i] = kEncInf; u.tmp_dirty[i] = 0; }
    if (kExact)   // (an aborted frame may have left bucket minima behind)
      for (int i = KH_TIDX; i < sh.x->x_hcap; i += NT) UX(x_bmin)[i] = 0xFFFFFFFFu;
    KhSync();
    KhDecodeStats st;
    DecodeOne<kLazy, kExact>(u, p, sh, &st);
    if (KH_TIDX == 0) {
      out[ui].stats = st;
      for (int i = 0; i < 4; i++) out[ui].sched[i] = sh->sched[i];
      out[ui].cand_mat = sh->cand_mat;
    }
    KhSync();
    if (st.status == 0) {
      if (kLazy) ExportSurvivors(u, p, pool, &out[ui], sh);
      else ExportLattice(u, p, pool, &out[ui], sh);
    }
    // The lattice pool, `out` and the completion list live in pinned HOST memory: the host
    // builds the utterance's canonical lattice and best path while this kernel decodes the
    // next ones.  Every wave's stores are complete behind KhSync(); thread 0 then releases
    // at system scope and appends the utterance to the list the host threads poll.
    KhSync();
    if (done_list != nullptr && KH_TIDX == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      const unsigned long long pos = __hip_atomic_fetch_add(&pool.used[3], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&done_list[pos], static_cast<int32_t>(ui), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    Stamp(u, sh, 9);
  }
  if (KH_TIDX == 0 && u.phase_cycles != nullptr)
    for (int i = 0; i < NPH; i++) u.phase_cycles[i] = sh->phase[i];
}


// ---------------------------------------------------------------- online decoding
// LatticeFasterOnlineDecoder (decoder/lattice-faster-online-decoder.{h,cc}): the
// same search, advanced a chunk of frames at a time.  A stream keeps its slot for
// the whole utterance; between launches the workgroup's LDS scalars live in
// SlotState.  One workgroup per job.
struct SlotState {
  int32_t tok_end, link_end, front_b, status, max_tokens_frame, tok_hw, gc_tok, gc_link;
  int32_t t, fb, fe;          // Run
  int32_t ok, finalized, conv_upto;
  int32_t surv_nt, surv_nl;   // (lazy schedule) survivor lists of FinalizeDecoding
  uint32_t x_hsize;           // (reference order) HashList::hash_size_ of the stream's decoder (:219-225: it never shrinks)
  int32_t pad_;
  long long arcs_expanded, tokens_created;
  KhDecodeStats stats;        // valid once finalized
};
enum { kJobInit = 0, kJobAdvance = 1, kJobFinalize = 2, kJobExport = 3 };
struct Job {
  int32_t slot, op;
  GP(const float) ll;         // kJobAdvance: matrix addressed by ABSOLUTE frame (chunk pointer - t * stride)
  int32_t ll_stride, n_frames;
};

__device__ void LoadState(const SlotState &S, Blk &sh, Run *run) {
  if (KH_TIDX == 0) {
    sh->tok_end = S.tok_end; sh->link_end = S.link_end; sh->front_b = S.front_b; sh->status = S.status;
    sh->max_tokens_frame = S.max_tokens_frame; sh->tok_hw = S.tok_hw; sh->gc_tok = S.gc_tok; sh->gc_link = S.gc_link;
    sh->arcs_expanded = S.arcs_expanded; sh->tokens_created = S.tokens_created; sh->conv_upto = S.conv_upto;
    sh->surv_nt = S.surv_nt; sh->surv_nl = S.surv_nl;
    sh->x_hsize = S.x_hsize;
  }
  run->t = S.t; run->fb = S.fb; run->fe = S.fe;
  run->cand = 0;   // (the online decoder does not report the counter)
  KhSync();
}
__device__ void SaveState(SlotState *S, Blk &sh, const Run &run, bool ok) {
  KhSync();
  if (KH_TIDX == 0) {
    S->tok_end = sh->tok_end; S->link_end = sh->link_end; S->front_b = sh->front_b; S->status = sh->status;
    S->max_tokens_frame = sh->max_tokens_frame; S->tok_hw = sh->tok_hw; S->gc_tok = sh->gc_tok; S->gc_link = sh->gc_link;
    S->arcs_expanded = sh->arcs_expanded; S->tokens_created = sh->tokens_created; S->conv_upto = sh->conv_upto;
    S->surv_nt = sh->surv_nt; S->surv_nl = sh->surv_nl;
    S->x_hsize = sh->x_hsize;
    S->t = run.t; S->fb = run.fb; S->fe = run.fe;
    S->ok = (ok && sh->status == 0) ? 1 : 0;
  }
}

// kExact: the reference's own iteration order (kh_online_decoder_set_reference_order; LatticeFasterOnlineDecoder::ProcessEmitting
// prunes against the same running cutoff in the same HashList order, lattice-faster-online-decoder.cc:864-951).
template <bool kExact>
__global__ void __launch_bounds__(NT)
#if KH_WG_PER_CU > 1
__attribute__((amdgpu_waves_per_eu(NT / 256 * KH_WG_PER_CU, NT / 256 * KH_WG_PER_CU)))
#endif
OnlineKernel(const Utt *__restrict__ slots, SlotState *__restrict__ states, const Job *__restrict__ jobs,
             UttOut *__restrict__ out, Pool pool, Params p, const UttX *slotsx) {
  __shared__ Shared shm;
  extern __shared__ float dyn_ll_row[];
  Blk sh;
  sh.p = (LdsShared *)&shm;
  sh.ll_row = (__attribute__((address_space(3))) float *)dyn_ll_row;
  InitCalls(sh, kExact);
  const Job job = jobs[blockIdx.x];
  sh.x = kExact ? (__attribute__((address_space(4))) const UttX *)(slotsx + job.slot) : nullptr;
  Utt u = slots[job.slot];
  Launder(u);
  Launder(p);
  SlotState *S = &states[job.slot];
  u.phase_cycles = (GP(long long))nullptr;
  if (KH_TIDX == 0) {
    for (int i = 0; i < NPH; i++) sh->phase[i] = 0;
    for (int i = 0; i < 4; i++) sh->orbuf[i] = 0;
    sh->hash_dirty = 0;
  }
  KhSync();
  Run run;
  if (job.op == kJobInit) {
    // arena invariants (a previous utterance of this stream may have left them dirty)
    const int hw = S->tok_hw;
    KhSync();
    for (int i = KH_TIDX; i < hw; i += NT) u.tok_cost[i] = kEncInf;
    for (uint32_t i = KH_TIDX; i <= u.hash_mask; i += NT) u.hash[i] = kEmpty;
    for (int i = KH_TIDX; i < u.tok_frame_cap; i += NT) { u.tmp_acc1[i] = kEncInf; u.tmp_dirty[i] = 0; }
    if (KH_TIDX == 0) sh->tok_hw = 0;
    KhSync();
    const bool ok = DecodeInit<kExact>(u, p, sh, &run);
    SaveState(S, sh, run, ok);
    if (KH_TIDX == 0) S->finalized = 0;
  } else if (job.op == kJobAdvance) {
    LoadState(*S, sh, &run);
    u.ll = job.ll;
    u.ll_stride = job.ll_stride;
    bool ok = S->ok != 0;
    if (ok) ok = p.lazy_prune ? DecodeFrames<true, kExact>(u, p, sh, &run, run.t + job.n_frames) : DecodeFrames<false, kExact>(u, p, sh, &run, run.t + job.n_frames);
    SaveState(S, sh, run, ok);
  } else if (job.op == kJobFinalize) {
    LoadState(*S, sh, &run);
    KhDecodeStats st;
    const bool ok = p.lazy_prune ? DecodeFinalize<true>(u, p, sh, run, S->ok != 0, &st) : DecodeFinalize<false>(u, p, sh, run, S->ok != 0, &st);
    SaveState(S, sh, run, ok);
    if (KH_TIDX == 0) { S->finalized = 1; S->stats = st; }
  } else {  // kJobExport: snapshot of the current lattice
    LoadState(*S, sh, &run);
    u.T = run.t;
    if (p.lazy_prune && !S->finalized && S->ok != 0 && run.t > 0) {
      // lazy schedule: nothing has been pruned since the last collection, and what is exported is what the arenas hold -
      // so the collection runs now (PruneActiveTokens at the current frame + compaction, as DecodeFrames does when an
      // arena runs low).  The snapshot is therefore pruned as of THIS frame, not as of the last multiple of
      // prune_interval; best paths and the final lattice do not depend on it (Params::lazy_prune).
      PruneActiveTokens(u, p, run.t, p.lattice_beam * p.prune_scale, sh);
      const bool okc = Compact(u, 0, run.t, p.keep_ac != 0, sh);
      if (KH_TIDX == 0) { sh->gc_tok = sh->tok_end; sh->gc_link = sh->link_end; sh->sched[0] += 1; }
      KhSync();
      run.fb = Uni(u.frame_b[run.t]);
      run.fe = Uni(u.frame_e[run.t]);
      SaveState(S, sh, run, okc);
      KhSync();
    }
    // FinalRelativeCost() before FinalizeDecoding (ComputeFinalCosts, lattice-faster-online-decoder.cc:860-900): over the
    // frontier tokens, best cost with the state's final cost - best cost; +inf when no token is in a final state.  What the
    // endpointing rules of online2/online-endpoint.cc test.
    float frc = INFINITY, fbest = INFINITY;
    if (!S->finalized) {
      float best_cost = INFINITY, best_with_final = INFINITY;
      for (int i = run.fb + KH_TIDX; i < run.fe; i += NT) {
        const float cost = Dec(LoadCostEnc(&u.tok_cost[i]));
        const float final_cost = __int_as_float(p.rec[u.tok_state[i]].w);
        best_cost = fminf(best_cost, cost);
        best_with_final = fminf(best_with_final, cost + final_cost);
      }
      best_cost = BlockMinF(best_cost, sh);
      best_with_final = BlockMinF(best_with_final, sh);
      frc = (best_cost == INFINITY && best_with_final == INFINITY) ? INFINITY : best_with_final - best_cost;
      fbest = best_with_final != INFINITY ? best_with_final : best_cost;
    }
    if (KH_TIDX == 0) {
      KhDecodeStats st = S->stats;
      if (!S->finalized) {
        st.num_frames = run.t; st.reached_final = 0; st.final_relative_cost = frc; st.final_best_cost = fbest;
        st.arcs_expanded = sh->arcs_expanded; st.tokens_created = sh->tokens_created;
        st.max_tokens_frame = sh->max_tokens_frame; st.num_tokens = sh->tok_end; st.num_links = sh->link_end;
      }
      st.status = sh->status;
      out[blockIdx.x].stats = st;
    }
    KhSync();
    if (p.lazy_prune && S->finalized) ExportSurvivors(u, p, pool, &out[blockIdx.x], sh);   // (FinalBackward's lists: the arenas are not pruned)
    else ExportLattice(u, p, pool, &out[blockIdx.x], sh);
  }
}


// ---------------------------------------------------------------- persistent serving kernel
// The same three jobs (InitDecoding / AdvanceDecoding / FinalizeDecoding) without a launch per chunk: one RESIDENT workgroup
// per stream waits on a control block in pinned host memory.  The host publishes `avail` = the number of frames whose scores
// are in the stream's score buffer (after the forward pass that wrote them has completed) and the workgroup decodes up to
// there - pruning at the reference's own frames (NumFramesDecoded() % prune_interval == 0, inside DecodeFrames) - at its own
// pace: a stream that prunes does not hold up the others, which is what a lockstep launch per chunk cannot avoid (the
// launch lasts as long as its slowest workgroup, and with streams out of phase some workgroup prunes in every launch).
// Commands (cmd_seq / cmd_op): InitDecoding has priority over pending frames, FinalizeDecoding runs after them.  After
// every action the slot's state is saved to memory (SlotState + arenas, agent-scope fence) before it is acknowledged, so
// an export job of OnlineKernel - launched by the host while this workgroup idles - sees it.  A workgroup leaves ONLY when
// *quit is set - by the host (kh_online_decoder_serve_stop), or by workgroup 0 once EVERY stream has been without work
// for idle_ticks (wall clock, 100 MHz; the streams' activity stamps in device memory): the kernel can never outlive its
// caller by more than that (a device-wide synchronisation waits that long at most), and the grid leaves as a whole.
// (Until round 5 every workgroup left on its own after idle_ticks: a stream that paused while the others kept decoding
// lost its workgroup, the host - which relaunches when the whole kernel has ended - never noticed, and the stream's next
// chunk or command waited for ever.  That is the intermittent stall of the round-4 serving legs.)
struct ServeCtl {
  int32_t avail, cmd_seq, cmd_op, pad0;     // host -> device
  int32_t ack_seq, decoded, ok, alive;      // device -> host
  // device -> host, diagnostics: what the stream's workgroup is doing (0 waiting, kActInit / kActAdvance / kActFinalize,
  // 9 = has left), the frame count it is advancing to, how many actions it has finished, the low word of the wall clock
  // (100 MHz) when it last started or finished one - what kh_online_decoder_serve_wait / _stop report when they time out
  int32_t hb_phase, hb_arg, hb_actions, hb_clock;
  int32_t hw;                               // device -> host: token slots of the stream's arena that hold something (InitDecoding resets them)
  int32_t pad1[3];                          // [0]: -DKH_SERVE_MARKERS progress word; [1]: device -> host, the kernel's status code behind ok == 0
};
static_assert(sizeof(ServeCtl) == 64, "one control block per 64-byte line");
// kCmdInitCleared: InitDecoding whose reset of the token arena the HOST has done (a fill kernel over the whole chip: the
// unpruned utterance of a lazy-schedule stream leaves ~70 MB to reset, 0.23 s for one workgroup - the 200 ms outliers of the
// round-4 chunk latencies were the first chunk of a slot's next utterance waiting for it)
enum { kCmdInit = 1, kCmdFinalize = 2, kCmdInitCleared = 3 };

template <class T>
__device__ __forceinline__ int32_t SysLoad(T p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
// ... and the polling side: the four words of a poll are fetched with relaxed loads (in flight together: one round trip to
// the host's memory instead of four, and no cache invalidation per word), followed by ONE acquire fence only when the poll
// found something to do (the scores the host published are read after it).
template <class T>
__device__ __forceinline__ int32_t SysLoadRelaxed(T p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// The host -> device words of a control block {avail, cmd_seq, cmd_op, pad} as ONE 16-byte read at system scope: one read
// request = one snapshot of the host's cache line, so a new cmd_seq is never seen with the operation or frame count of an
// earlier command (the host stores avail, cmd_op, cmd_seq in that order; separate relaxed reads may be reordered on the bus).
typedef int KhI4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ KhI4 SysLoad16(const void *p) {
  KhI4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <class T>
__device__ __forceinline__ void SysStore(T p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
// A store to the host's control block that orders nothing: diagnostics and fields the host reads only after the one
// RELEASE store that ends an action.  A system-scope release is a write-back of the XCD's whole L2 (buffer_wbl2 sc0 sc1) -
// with 32 streams decoding per XCD that is everybody's dirty lines; round 5 had grown from three to ten of them per chunk
// (the heart-beat fields), round 6 issues exactly one.
template <class T>
__device__ __forceinline__ void SysStoreRelaxed(T p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// kExact: the streams decode in the reference's own iteration order (OnlineKernel<true>'s routines; the slots' temporaries in slotsx).
template <bool kExact>
__global__ void __launch_bounds__(NT)
#if KH_WG_PER_CU > 1
__attribute__((amdgpu_waves_per_eu(NT / 256 * KH_WG_PER_CU, NT / 256 * KH_WG_PER_CU)))
#endif
ServeKernel(const Utt *__restrict__ slots, SlotState *__restrict__ states, ServeCtl *ctl, int32_t *quit,
            const float *ll_base, long long ll_rows_per_stream, int ll_stride, Params p, long long idle_ticks,
            long long *act_clock /* [gridDim.x] wall clock of every stream's last activity */, const UttX *slotsx) {
  __shared__ Shared shm;
  extern __shared__ float dyn_ll_row[];
  Blk sh;
  sh.p = (LdsShared *)&shm;
  sh.ll_row = (__attribute__((address_space(3))) float *)dyn_ll_row;
  InitCalls(sh, kExact);
  const int s = blockIdx.x;
  sh.x = kExact ? (__attribute__((address_space(4))) const UttX *)(slotsx + s) : nullptr;
  Utt u = slots[s];
  Launder(u);
  Launder(p);
  SlotState *S = &states[s];
  ServeCtl *c = &ctl[s];
  u.phase_cycles = (GP(long long))nullptr;
  u.ll = (GP(const float))(ll_base + static_cast<size_t>(s) * ll_rows_per_stream * ll_stride);
  u.ll_stride = ll_stride;
  if (KH_TIDX == 0) {
    for (int i = 0; i < NPH; i++) sh->phase[i] = 0;
    for (int i = 0; i < 4; i++) sh->orbuf[i] = 0;
    sh->hash_dirty = 0;
  }
  KhSync();
  Run run;
  LoadState(*S, sh, &run);
  bool ok = S->ok != 0;
  bool fin = S->finalized != 0;
  int acked = c->ack_seq;   // (the host set cmd_seq = ack_seq before the launch)
  long long t_idle = wall_clock64(), t_scan = t_idle;
  int n_actions = 0, idle_polls = 0;
  if (KH_TIDX == 0) {
    __hip_atomic_store(&act_clock[s], t_idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SysStore(&c->hb_phase, 0);
    SysStore(&c->hb_clock, static_cast<int32_t>(t_idle));
    SysStore(&c->alive, 1);
  }
  for (;;) {
    if (KH_TIDX == 0) {
      int act = 0, arg = 0;
      const KhI4 w = SysLoad16(c);      // {avail, cmd_seq, cmd_op, pad0}
      const int av = w.x, seq = w.y, op = w.z;
      const int q = SysLoadRelaxed(quit);
      if (seq != acked && (op == kCmdInit || op == kCmdInitCleared)) { act = 1; arg = op == kCmdInitCleared ? 1 : 0; }
      else if (!fin && ok && av > run.t) { act = 2; arg = av; }
      else if (seq != acked && op == kCmdFinalize) act = 3;
      else if (seq != acked) act = 5;   // an unknown command: acknowledged, nothing done
      else if (q != 0) act = 4;
      else if (s == 0) {
        // the grid's idle decision: this workgroup has been without work for idle_ticks - has every other one?  (looked at
        // at most four times per idle_ticks; a stream in the middle of a long action carries an old stamp: it finishes the
        // action, acknowledges it and leaves with the others, and the host's next call launches the grid again)
        const long long now = wall_clock64();
        if (now - t_idle > idle_ticks && now - t_scan > (idle_ticks >> 2)) {
          t_scan = now;
          bool all_idle = true;
          for (int k = 1; k < static_cast<int>(gridDim.x) && all_idle; k++)
            all_idle = now - __hip_atomic_load(&act_clock[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > idle_ticks;
          if (all_idle) SysStore(quit, 2);
        }
      }
      if (act != 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: what the host wrote before it published
      sh->bcast_i[0] = act;
      sh->bcast_i[1] = arg;
      sh->bcast_i[2] = seq;
      if (act != 0 && act != 4) {
        SysStoreRelaxed(&c->hb_arg, arg);
        SysStoreRelaxed(&c->hb_clock, static_cast<int32_t>(wall_clock64()));
        SysStoreRelaxed(&c->hb_phase, act);
      }
    }
    KhSync();
    const int act = Uni(sh->bcast_i[0]), arg = Uni(sh->bcast_i[1]), seq = Uni(sh->bcast_i[2]);
    KhSync();
    if (act == 0) {
      // Back off while nothing happens: every poll is a read of the host's memory across the bus, 256 resident workgroups
      // polling every 3.4 us are 75 M reads a second in front of the packets the command processor fetches for the host's
      // own kernels (the forward pass of the next chunk): a stream that has polled in vain a few times waits 14, then 27 us
      // between polls (a chunk is 1.5 - 4 ms of work away; the first polls after an action stay at full rate).
#ifndef KH_SERVE_NO_BACKOFF
      idle_polls++;
      const int reps = idle_polls < 8 ? 1 : (idle_polls < 32 ? 4 : 8);
#else
      const int reps = 1;
#endif
      for (int r = 0; r < reps; r++) __builtin_amdgcn_s_sleep(127);
      continue;
    }
    idle_polls = 0;
    if (act == 4) break;
    if (act == 1) {          // InitDecoding (the body of OnlineKernel's kJobInit)
      const int hw = arg != 0 ? 0 : S->tok_hw;   // (arg: the host has reset the token arena)
      KhSync();
      for (int i = KH_TIDX; i < hw; i += NT) u.tok_cost[i] = kEncInf;
      for (uint32_t i = KH_TIDX; i <= u.hash_mask; i += NT) u.hash[i] = kEmpty;
      for (int i = KH_TIDX; i < u.tok_frame_cap; i += NT) { u.tmp_acc1[i] = kEncInf; u.tmp_dirty[i] = 0; }
      if (KH_TIDX == 0) sh->tok_hw = 0;
      KhSync();
      ok = DecodeInit<kExact>(u, p, sh, &run);
      SaveState(S, sh, run, ok);
      if (KH_TIDX == 0) S->finalized = 0;
      fin = false;
      ok = ok && Uni(sh->status) == 0;
    } else if (act == 2) {   // AdvanceDecoding up to the frames the host has published
      LoadState(*S, sh, &run);   // (an export job may have collected the slot's garbage in between: lazy schedule)
      ok = ok && S->ok != 0;
      if (ok) ok = p.lazy_prune ? DecodeFrames<true, kExact>(u, p, sh, &run, arg) : DecodeFrames<false, kExact>(u, p, sh, &run, arg);
      SaveState(S, sh, run, ok);
      ok = ok && Uni(sh->status) == 0;
    } else if (act == 3) {   // FinalizeDecoding
      KhDecodeStats st;
      LoadState(*S, sh, &run);
      ok = p.lazy_prune ? DecodeFinalize<true>(u, p, sh, run, ok, &st) : DecodeFinalize<false>(u, p, sh, run, ok, &st);
      SaveState(S, sh, run, ok);
      if (KH_TIDX == 0) { S->finalized = 1; S->stats = st; }
      fin = true;
    }
    __threadfence();   // the slot's arenas and SlotState are in memory before the acknowledgement
    KhSync();
    n_actions++;
    t_idle = wall_clock64();
    if (KH_TIDX == 0) {
      __hip_atomic_store(&act_clock[s], t_idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (the __threadfence + barrier above have put the slot's arenas and SlotState in memory: of the stores below only the
      // LAST one - what the host waits for - needs to order the others)
      SysStoreRelaxed(&c->pad1[1], sh->status);   // (the code behind ok == 0)
      SysStoreRelaxed(&c->ok, ok ? 1 : 0);
      SysStoreRelaxed(&c->hw, sh->tok_hw);
      SysStoreRelaxed(&c->hb_actions, n_actions);
      SysStoreRelaxed(&c->hb_clock, static_cast<int32_t>(t_idle));
      SysStoreRelaxed(&c->hb_phase, 0);
      if (act != 2) {
        SysStoreRelaxed(&c->decoded, run.t);
        SysStore(&c->ack_seq, seq);
      } else {
        SysStore(&c->decoded, run.t);
      }
    }
    if (act != 2) acked = seq;
  }
  KhSync();
  if (KH_TIDX == 0) {
    SysStore(&c->hb_phase, 9);
    SysStore(&c->alive, 0);
  }
}

__global__ void FillU32(uint32_t *p, size_t n, uint32_t v) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + KH_TIDX; i < n;
       i += static_cast<size_t>(gridDim.x) * blockDim.x)
    p[i] = v;
}

}  // namespace

// ================================================================ host side
struct KhDecoder {
  const KhFst *fst = nullptr;
  KhDecoderConfig cfg;
  int max_batch = 0, max_frames = 0;
  int tok_frame_cap = 0, link_frame_cap = 0, expected_tokens = 0;
  int win_tok = 0, win_link = 0;   // tokens / links per frame the compaction window is sized for (averages, not caps)
  int max_slots = 0;
  // slot arenas (one set per persistent workgroup)
  void *slab = nullptr;
  size_t slab_bytes = 0;
  int slab_slots = 0, slab_T = 0, slab_scale = 1;
  int lazy = 0, alloc_link_a = 1;         // Params::lazy_prune / keep_ac of the calls this decoder serves (kh_decoder_decode sets them)
  int slab_lazy = 0, slab_link_a = 1;     // ... and what the slab was carved for
  int exact = 1, slab_exact = 0;          // kh_decoder_set_reference_order (default since round 6: the reference's own order); whether the slab holds the exact-order temporaries
  int rec_order_ids = -1;            // what the second word of the arcs in `rec` holds: 0 output labels, 1 state ids (ArcPdfKernel), -1 not built
  std::vector<Utt> h_slots;
  Utt *d_slots = nullptr;
  std::vector<UttX> h_slotsx;      // exact reference order: the slots' temporaries (Utt::x points into d_slotsx)
  UttX *d_slotsx = nullptr;
  UttIn *d_in = nullptr;
  UttOut *d_out = nullptr;
  unsigned long long *d_used = nullptr;
  long long *d_phase = nullptr;
  int4 *rec = nullptr;             // BuildArcPdf: the state records with the pdf of every arc
  int *d_bad = nullptr;            // BuildArcPdf: {flag, transition-id, pdf} of a map entry outside the score matrix

  // lattice pool
  void *pool_slab = nullptr;
  size_t pool_bytes = 0;
  Pool pool;
  struct HostPool {
    std::vector<int32_t> t_frame, t_state, l_src, l_dst, l_il, l_ol;
    std::vector<float> l_g, l_a;
  };
  std::vector<HostPool> rounds;     // online snapshots: host copy of the device pool
  std::vector<int32_t> h_round;     // utterance -> entry of `rounds` holding its lattice; -1 = the pinned pool below
  // Offline decoding: the kernel exports finished lattices straight into pinned host memory
  // and the host builds them while the kernel is still running (kh_decoder_decode).
  void *hp_slab = nullptr;          // pinned: pool arrays
  size_t hp_bytes = 0;
  Pool hpool;                       // device-side pointers into hp_slab (+ used counters in device memory)
  struct PoolView {
    const int32_t *t_frame, *t_state, *l_src, *l_dst, *l_il, *l_ol;
    const float *l_g, *l_a;
  } hview;                          // host-side pointers into hp_slab
  UttOut *h_out_pinned = nullptr;   // [max_batch] pinned
  int32_t *h_done = nullptr;        // [max_batch] pinned completion list
  // bump allocators of the worker threads for the canonical lattices (chunks are kept across calls)
  struct Arena {
    std::vector<std::unique_ptr<char[]>> chunks;
    std::vector<size_t> sizes;
    size_t cur = 0, used = 0;
    void Reset() { cur = 0; used = 0; }
    void *Take(size_t bytes) {
      bytes = (bytes + 63) & ~static_cast<size_t>(63);
      while (cur < chunks.size() && used + bytes > sizes[cur]) { cur++; used = 0; }
      if (cur == chunks.size()) {
        const size_t sz = std::max<size_t>(bytes, 32u << 20);
        chunks.emplace_back(new char[sz]);
        sizes.push_back(sz);
        used = 0;
      }
      void *p = chunks[cur].get() + used;
      used += bytes;
      return p;
    }
  };
  std::vector<Arena> arenas;
  std::vector<UttOut> h_out;
  std::vector<int32_t> h_T;
  std::vector<int32_t> order;  // queue position -> utterance
  int n_utts = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_kernel_ms = 0.f;
  float last_host_tail_ms = 0.f;   // wall time the host threads needed after the stream was idle (summed over the launches)
  int slot_limit = std::numeric_limits<int>::max();  // slots that fit in memory (found by a failed allocation)
  // canonical lattices, built lazily per utterance
  // Array of a canonical lattice: its own vector, or (batch post-pass, kh_decoder_prepare) a
  // slice of the decoder's batch store.  One mmap + munmap per vector per utterance serialises
  // the host threads on the process's mm lock (0.7 s per 2620-utterance step before).
  template <class T>
  struct Buf {
    T *p = nullptr;
    size_t n = 0;
    bool bound = false;
    std::vector<T> own;
    void bind(T *ext, size_t count) { p = ext; n = count; bound = true; }
    void resize(size_t count) {
      if (bound && count == n) return;
      bound = false;
      own.resize(count);
      p = own.data();
      n = count;
    }
    void assign(size_t count, T v) {
      resize(count);
      for (size_t i = 0; i < count; i++) p[i] = v;
    }
    size_t size() const { return n; }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
  };
  struct Lat {
    bool built = false;
    Buf<int32_t> state_frame, state_hclg;
    Buf<float> state_final;
    Buf<int32_t> arc_src, arc_dst, arc_il, arc_ol;
    Buf<float> arc_g, arc_a;
    // best path (GetBestPath), cached
    int bp_rc = 1;  // 1 = not computed yet
    std::vector<int32_t> bp_ali, bp_words;
    float bp_graph = 0.f, bp_acoustic = 0.f;
  };
  std::vector<Lat> lats;
  // determinization behind the decoder (kh_decoder_set_determinize): the CompactLattice of every utterance
  bool det_enable = false;
  void (*after_launch)(void *) = nullptr;   // kh_decoder_set_after_launch
  void *after_launch_arg = nullptr;
  double det_beam = 0.0;
  float det_delta = 0.0f;
  int64_t det_max_mem = 0;
  int det_phone = 0, det_word = 1, det_minimize = 0;
  std::vector<int32_t> det_tid_phone;
  std::vector<KhCompactLattice *> clats;
  void FreeClats() {
    for (KhCompactLattice *c : clats) if (c) kh_compact_lattice_free(c);
    clats.clear();
  }
  // batch store of kh_decoder_prepare (kept across calls: the pages stay mapped)
  std::vector<int32_t> st_i[6];   // state_frame, state_hclg, arc_src, arc_dst, arc_il, arc_ol
  std::vector<float> st_f[3];     // state_final, arc_g, arc_a
};

// LatticeFasterOnlineDecoder for num_streams concurrent utterances: stream i owns
// slot i of `base` for the whole utterance.
struct KhOnlineDecoder {
  KhDecoder *base = nullptr;
  int num_streams = 0, max_frames = 0;
  SlotState *d_states = nullptr;
  Job *d_jobs = nullptr;
  std::vector<int32_t> frames;       // NumFramesDecoded() per stream
  std::vector<char> inited, finalized;
  std::vector<long long> lat_key;    // what base->lats[stream] was built from (-1: nothing)
  // kh_online_decoder_set_pdf_map: the (map, columns) pair the decoder's arc records were last built for
  const int32_t *pinned_map = nullptr;
  int pinned_cols = 0;
  bool pinned = false;
  // persistent serving kernel (kh_online_decoder_serve_*)
  ServeCtl *serve_ctl = nullptr;          // pinned host memory, one 64-byte block per stream
  int32_t *serve_quit = nullptr;          // pinned
  hipStream_t serve_stream = nullptr;
  bool serve_launched = false;            // a kernel has been launched and not yet seen to have ended
  long long *d_serve_act = nullptr;       // device: wall clock of every stream's last activity (the grid's idle decision)
  long long serve_relaunches = 0;         // how often the kernel was launched again after it had left
  const float *serve_ll = nullptr;
  long long serve_rows = 0;
  int serve_stride = 0;
  const int32_t *serve_map = nullptr;
  std::vector<int32_t> serve_seq;         // last command sequence number issued per stream
};

namespace {

// KhDecodeStats::status / the codes the kernels leave in Shared::status
const char *StatusText(int code) {
  switch (code) {
    case 0: return "ok";
    case 6: return "the lattice did not fit the pool";
    case 7: return "the survivor lists are full";
    case 8: case 9: return "reference order: the list order could not be built";
    case 10: return "a frame's epsilon links are not a DAG - internal inconsistency";
    default: return "capacity: a token / link arena or a per-frame cap overflowed";
  }
}

// CPUs this process may use: the cgroup's cpu.max quota / period (v2; cpu.cfs_quota_us / cpu.cfs_period_us in v1), else
// the hardware concurrency.  A container with 256 visible cores and a 16-CPU quota runs 16 threads well and 160 badly.
int HostCpuQuota() {
  static const int quota = [] {
    int hw = static_cast<int>(std::thread::hardware_concurrency());
    if (hw <= 0) hw = 1;
    long long q = -1, per = -1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char buf[64] = {0};
      if (fscanf(f, "%63s %lld", buf, &per) == 2 && strcmp(buf, "max") != 0) q = atoll(buf);
      fclose(f);
    } else {
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &q) != 1) q = -1; fclose(g); }
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &per) != 1) per = -1; fclose(g); }
    }
    if (q > 0 && per > 0) hw = std::max(1, std::min<int>(hw, static_cast<int>((q + per - 1) / per)));
    return hw;
  }();
  return quota;
}

size_t Align(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

struct Carver {
  char *base;
  size_t off = 0;
  template <class T>
  GP(T) Take(size_t n) {
    T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
    off += Align(n * sizeof(T));
    return (GP(T))p;
  }
};

// Token / link capacity of a slot's arenas.
struct ArenaCaps {
  long long tok = 0, link = 0;
};
// every arena array holds 4-byte elements addressed with a 32-bit byte offset (Arr<T>): < 2^30 slots
constexpr long long kMaxArenaSlots = (1ll << 30) - 1;

// The capacity the windowed schedule (PruneActiveTokens + compaction every prune_interval frames) needs:
// the compaction window at its average size + lattice density for the frames behind it.
ArenaCaps WindowedCaps(int T, int tok_frame_cap, int link_frame_cap, int win_tok, int win_link, int prune_interval) {
  // window: 2 * max(prune_interval, 25) frames + one interval of new frames + frontier;
  // stable part: lattice density
  const long long win_frames = std::min<long long>((KH_COMPACT_EVERY + 1ll) * std::max(prune_interval, 25) + 1ll * KH_COMPACT_EVERY * prune_interval + 3, T + 2);
  long long per_frame = 256;
  if (const char *e = getenv("KH_DECODER_STABLE_TOKENS_PER_FRAME")) per_frame = atoll(e);
  const long long stable_tok = std::max<long long>(65536, per_frame * (T + 2));
  // the window holds win_frames frames of AVERAGE size (win_tok / win_link per frame) plus one
  // frame of the per-frame caps; an utterance that needs more reports an overflow and is
  // decoded again with larger arenas (kh_decoder_decode)
  ArenaCaps c;
  c.tok = std::min(kMaxArenaSlots, stable_tok + win_frames * win_tok + tok_frame_cap);
  c.link = std::min(kMaxArenaSlots, 3 * stable_tok + win_frames * win_link + link_frame_cap);
  return c;
}

// The capacity the lazy schedule would like: the whole utterance unpruned at the window's average frame size
// (links: twice the tokens - a frame materialises ~1.6 candidates per token it creates), never less than the
// windowed one.  EnsureSlots scales it down to the memory there is; a slot that fills up collects its garbage.
ArenaCaps LazyCaps(const ArenaCaps &floor, int T, int tok_frame_cap, int link_frame_cap, int win_tok) {
  ArenaCaps c;
  c.tok = std::min(kMaxArenaSlots, std::max(floor.tok, (T + 2ll) * win_tok + tok_frame_cap));
  c.link = std::min(kMaxArenaSlots, std::max(floor.link, 2 * (T + 2ll) * win_tok + 2ll * link_frame_cap));
  return c;
}

// Arena set of one slot, sized for utterances of up to T frames.
void CarveSlot(Carver &c, Utt &u, int T, int tok_frame_cap, int link_frame_cap, const ArenaCaps &caps, bool link_a,
               float hash_ratio, int expected_tokens, bool exact, UttX *xo = nullptr, int scale = 1) {
  u.T = T;
  u.tok_frame_cap = tok_frame_cap;
  u.link_frame_cap = link_frame_cap;
  u.window_cap = static_cast<int32_t>(caps.tok);
  u.tok_cap = static_cast<int32_t>(caps.tok);
  u.link_cap = static_cast<int32_t>(caps.link);
  const size_t nt = u.tok_cap, nl = u.link_cap;
  u.tok_state = c.Take<int32_t>(nt);
  u.tok_cost = c.Take<uint32_t>(nt);
  u.tok_extra = c.Take<float>(nt);
  u.tmp_remap = c.Take<int32_t>(nt);
  u.link_dst = c.Take<int32_t>(nl);
  u.link_src = c.Take<int32_t>(nl);
  u.link_arc = c.Take<int32_t>(nl);
  u.link_k = c.Take<float>(nl);
  u.link_a = c.Take<float>(link_a ? nl : 0);   // (offline decoding recomputes the acoustic costs at export: Params::keep_ac)
  u.frame_b = c.Take<int32_t>(T + 2);
  u.frame_e = c.Take<int32_t>(T + 2);
  u.feps_b = c.Take<int32_t>(T + 2);
  u.feps_e = c.Take<int32_t>(T + 2);
  u.femit_b = c.Take<int32_t>(T + 2);
  u.femit_e = c.Take<int32_t>(T + 2);
  u.cost_offset = c.Take<float>(T + 1);
  u.must_links = c.Take<uint8_t>(T + 2);
  u.must_toks = c.Take<uint8_t>(T + 2);
  u.tmp_slot = c.Take<int32_t>(tok_frame_cap);
  u.tmp_dirty = c.Take<int32_t>(tok_frame_cap);
  u.tmp_work0 = c.Take<int32_t>(tok_frame_cap);
  u.tmp_work1 = c.Take<int32_t>(tok_frame_cap);
  u.tmp_epslist = c.Take<int32_t>(tok_frame_cap);
  u.tmp_f0 = c.Take<float>(tok_frame_cap);
  u.tmp_acc0 = c.Take<uint32_t>(tok_frame_cap);
  u.tmp_acc1 = c.Take<uint32_t>(tok_frame_cap);
  // survivor lists of FinalBackward: ~4 x the lattice density the recipe's options give (17 states / 27 arcs per frame);
  // an utterance that needs more is decoded again with everything doubled (tok_frame_cap scales with the retry)
  u.surv_tok_cap = static_cast<int32_t>(std::min<long long>(caps.tok, 64ll * scale * (T + 2) + tok_frame_cap));
  u.surv_link_cap = static_cast<int32_t>(std::min<long long>(caps.link, 128ll * scale * (T + 2) + 2ll * tok_frame_cap));
  u.surv_tok = c.Take<int32_t>(2 * static_cast<size_t>(u.surv_tok_cap));
  u.surv_link = c.Take<int32_t>(2 * static_cast<size_t>(u.surv_link_cap));
  // hash_ratio x the tokens a frame is expected to hold (lattice-faster-decoder.cc:193-199
  // resizes to hash_ratio x the previous frame's count); never fewer entries than a
  // frame may hold.  A smaller table keeps more of it in L2.
  size_t hs = 1;
  while (hs < static_cast<size_t>(hash_ratio * expected_tokens) || hs <= static_cast<size_t>(tok_frame_cap)) hs <<= 1;
  u.hash_mask = static_cast<uint32_t>(hs - 1);
  u.hash = c.Take<unsigned long long>(hs);
  // exact reference order: per-frame temporaries (8 MB per slot at the default caps)
  {
    UttX x;
    const size_t tf = exact ? tok_frame_cap : 0, lf = exact ? link_frame_cap : 0;
    // the reference's table has at most hash_ratio x (tokens of a frame) buckets, 1000 to begin with (:37, :219-225)
    x.x_hcap = exact ? static_cast<int32_t>(std::max<double>(1000.0, static_cast<double>(hash_ratio) * tok_frame_cap) + 16) : 0;
    x.x_pos = c.Take<int32_t>(tf);
    x.x_m = c.Take<uint32_t>(tf);
    x.x_c = c.Take<int32_t>(tf);
    x.x_q = c.Take<uint32_t>(tf);
    x.x_bkt = c.Take<int32_t>(tf);
    x.x_epsidx = c.Take<int32_t>(tf);
    x.x_nl0 = c.Take<int32_t>(tf);
    x.x_nl1 = c.Take<int32_t>(tf);
    x.x_ncost = c.Take<float>(tf);
    x.x_ord = c.Take<int32_t>(lf);
    x.x_lw = c.Take<float>(lf);
    x.x_stack = c.Take<int32_t>(lf);
    x.x_bmin = c.Take<uint32_t>(static_cast<size_t>(x.x_hcap));
    x.x_key0 = c.Take<unsigned long long>(tf);
    x.x_key1 = c.Take<unsigned long long>(tf);
    x.x_val0 = c.Take<int32_t>(tf);
    x.x_val1 = c.Take<int32_t>(tf);
    x.x_h = c.Take<int32_t>(tf);
    x.x_inb = c.Take<int32_t>(tf);
    x.x_c0e = c.Take<uint32_t>(tf);
    x.x_csid = c.Take<int32_t>(lf);
    x.x_rec0 = Arr<const KhInt4>((GP(const KhInt4))nullptr);
    if (xo) *xo = x;
  }
  u.ll = (GP(const float))nullptr;
  u.ll_stride = 0;
  u.phase_cycles = (GP(long long))nullptr;
}

// canonical lattice of one utterance from the host copy of the pool (GetRawLattice
// :109-191 with use_final_probs = true after FinalizeDecoding)
int BuildLattice(KhDecoder *d, int ui) {
  KhDecoder::Lat &L = d->lats[ui];
  if (L.built) return KH_OK;
  const UttOut &o = d->h_out[ui];
  const KhDecodeStats &st = o.stats;
  if (st.status != 0) {
    SetError("utterance %d: decoder failed (code %d: %s)", ui, st.status, StatusText(st.status));
    return KH_ECAPACITY;
  }
  const int T = d->h_T[ui];
  const size_t n = o.n_tok, m = o.n_link;
  KhDecoder::PoolView hp;
  if (d->h_round[ui] < 0) {
    hp = d->hview;
  } else {
    const KhDecoder::HostPool &r = d->rounds[d->h_round[ui]];
    hp = KhDecoder::PoolView{r.t_frame.data(), r.t_state.data(), r.l_src.data(), r.l_dst.data(), r.l_il.data(),
                             r.l_ol.data(), r.l_g.data(), r.l_a.data()};
  }
  const int32_t *tf = hp.t_frame + o.tok_off, *ts = hp.t_state + o.tok_off;
  // (scratch vectors are per thread and only grow: no allocation per utterance)
  static thread_local std::vector<int32_t> ord, newidx, bucket;
  // canonical order (frame, HCLG state) with the start token first: lattice state 0 is the
  // start state (the reference gets that from TopSortTokens :839-914; ComputeBestPath and
  // the lattice writers rely on it), also when the start state has an epsilon arc to a
  // lower-numbered state.  Counting sort by frame (the kernel exports a frame's tokens together), then the
  // handful of tokens of a frame by state: two comparison sorts over the whole utterance were 2.5 ms of CPU per
  // utterance - as much as its determinization - on boxes whose CPU quota is what bounds the end-to-end step.
  const int32_t start = d->fst->start_state;
  int32_t max_f = 0;
  for (size_t k = 0; k < n; k++) {
    if (tf[k] < 0) { SetError("utterance %d: exported token %zu has frame %d", ui, k, tf[k]); return KH_ESTATE; }
    max_f = std::max(max_f, tf[k]);
  }
  bucket.assign(static_cast<size_t>(max_f) + 2, 0);
  for (size_t k = 0; k < n; k++) bucket[tf[k] + 1]++;
  for (int f = 0; f <= max_f; f++) bucket[f + 1] += bucket[f];
  ord.resize(n);
  {
    static thread_local std::vector<int32_t> fill;
    fill.assign(bucket.begin(), bucket.end() - 1);
    for (size_t k = 0; k < n; k++) ord[fill[tf[k]]++] = static_cast<int32_t>(k);
  }
  for (int f = 0; f <= max_f; f++) {
    int32_t *b = ord.data() + bucket[f], *e = ord.data() + bucket[f + 1];
    auto less = [&](int32_t x, int32_t y) {
      if (f == 0) {
        const bool xs = ts[x] != start, ys = ts[y] != start;
        if (xs != ys) return xs < ys;
      }
      return ts[x] < ts[y];
    };
    if (e - b <= 24) {
      for (int32_t *q = b + 1; q < e; q++) {
        const int32_t v = *q;
        int32_t *r = q;
        for (; r > b && less(v, r[-1]); r--) *r = r[-1];
        *r = v;
      }
    } else {
      std::sort(b, e, less);
    }
  }
  newidx.resize(n);
  for (size_t k = 0; k < n; k++) newidx[ord[k]] = static_cast<int32_t>(k);
  const float inf = std::numeric_limits<float>::infinity();
  L.state_frame.resize(n);
  L.state_hclg.resize(n);
  L.state_final.assign(n, inf);
  const bool have_final = st.reached_final != 0;
  for (size_t k = 0; k < n; k++) {
    const int32_t i = ord[k];
    L.state_frame[k] = tf[i];
    L.state_hclg[k] = ts[i];
    if (tf[i] == T) {  // :177-186
      if (have_final) {
        const float fc = d->fst->final_host[ts[i]];
        if (fc != inf) L.state_final[k] = fc;
      } else {
        L.state_final[k] = 0.0f;
      }
    }
  }
  // arcs in the order (src, ilabel, olabel, dst, graph cost, acoustic cost): counting sort by source state, then the few
  // arcs of a state among themselves
  struct A { int32_t src, il, ol, dst; float g, a; };
  static thread_local std::vector<A> arcs;
  static thread_local std::vector<int32_t> aoff;
  arcs.resize(m);
  const int32_t *ls = hp.l_src + o.link_off, *ld = hp.l_dst + o.link_off,
                *li = hp.l_il + o.link_off, *lo = hp.l_ol + o.link_off;
  const float *lg = hp.l_g + o.link_off, *la = hp.l_a + o.link_off;
  aoff.assign(n + 1, 0);
  for (size_t j = 0; j < m; j++) {
    if (ls[j] < 0 || static_cast<size_t>(ls[j]) >= n || ld[j] < 0 || static_cast<size_t>(ld[j]) >= n) {
      SetError("utterance %d: exported link %zu points outside the %zu tokens", ui, j, n);
      return KH_ESTATE;
    }
    aoff[newidx[ls[j]] + 1]++;
  }
  for (size_t k = 0; k < n; k++) aoff[k + 1] += aoff[k];
  {
    static thread_local std::vector<int32_t> fill;
    fill.assign(aoff.begin(), aoff.end() - 1);
    for (size_t j = 0; j < m; j++) {
      const int32_t sidx = newidx[ls[j]];
      arcs[fill[sidx]++] = A{sidx, li[j], lo[j], newidx[ld[j]], lg[j], la[j]};
    }
  }
  auto arc_less = [](const A &x, const A &y) {
    if (x.il != y.il) return x.il < y.il;
    if (x.ol != y.ol) return x.ol < y.ol;
    if (x.dst != y.dst) return x.dst < y.dst;
    if (x.g != y.g) return x.g < y.g;
    return x.a < y.a;
  };
  for (size_t k = 0; k < n; k++) {
    A *b = arcs.data() + aoff[k], *e = arcs.data() + aoff[k + 1];
    if (e - b <= 24) {
      for (A *q = b + 1; q < e; q++) {
        const A v = *q;
        A *r = q;
        for (; r > b && arc_less(v, r[-1]); r--) *r = r[-1];
        *r = v;
      }
    } else {
      std::sort(b, e, arc_less);
    }
  }
  L.arc_src.resize(m); L.arc_dst.resize(m); L.arc_il.resize(m);
  L.arc_ol.resize(m); L.arc_g.resize(m); L.arc_a.resize(m);
  for (size_t j = 0; j < m; j++) {
    L.arc_src[j] = arcs[j].src; L.arc_dst[j] = arcs[j].dst; L.arc_il[j] = arcs[j].il;
    L.arc_ol[j] = arcs[j].ol; L.arc_g[j] = arcs[j].g; L.arc_a[j] = arcs[j].a;
  }
  L.built = true;
  return KH_OK;
}

struct LatWeight { float v1, v2; };
// fstext/lattice-weight.h:297-312
inline int Compare(const LatWeight &w1, const LatWeight &w2) {
  const float f1 = w1.v1 + w1.v2, f2 = w2.v1 + w2.v2;
  if (f1 < f2) return 1;
  else if (f1 > f2) return -1;
  else if (w1.v1 < w2.v1) return 1;
  else if (w1.v1 > w2.v1) return -1;
  else return 0;
}

// Best path of utterance `utt` straight from the exported pool arrays, cached in the Lat: the result of ComputeBestPath
// below (same LatticeWeight arithmetic, same tie rule) without building the canonical lattice first - two counting sorts
// and one relaxation per link instead of the canonical sort of ~12 k states / ~20 k arcs and Bellman-Ford sweeps over all
// of them (2 ms of CPU per utterance, which a rank with two host threads does not have: DESIGN.md section 5).
// The raw lattice is layered: an emitting link goes from frame f to f + 1, an epsilon link stays inside its frame.  So the
// links are bucketed by (destination frame, emitting before epsilon) and relaxed frame by frame, the epsilon links of a
// frame to their fixed point.  Tie rule: ComputeBestPath prefers the smaller CANONICAL arc index among the arcs that reach
// a state with Compare-equal weights; canonical arcs are ordered by (source state's canonical index = (frame, start token
// first, HCLG state), ilabel, olabel, destination, graph cost, acoustic cost), which is compared on the keys here.
// Returns 1 if the pool's layout is not the one expected (the caller then takes the canonical route).
int ComputeBestPathLean(KhDecoder *d, int utt) {
  KhDecoder::Lat &L = d->lats[utt];
  const UttOut &o = d->h_out[utt];
  if (o.stats.status != 0) {
    SetError("utterance %d: decoder failed (code %d: %s)", utt, o.stats.status, StatusText(o.stats.status));
    return KH_ECAPACITY;
  }
  if (d->h_round[utt] >= 0) return 1;   // (online snapshots arrive built)
  const KhDecoder::PoolView hp = d->hview;
  const int T = d->h_T[utt];
  const int n = static_cast<int>(o.n_tok), m = static_cast<int>(o.n_link);
  if (n == 0) {
    SetError("GetBestPath: empty lattice for utterance %d", utt);
    return L.bp_rc = KH_ESTATE;
  }
  const int32_t *tf = hp.t_frame + o.tok_off, *ts = hp.t_state + o.tok_off;
  const int32_t *ls = hp.l_src + o.link_off, *ld = hp.l_dst + o.link_off, *li = hp.l_il + o.link_off, *lo = hp.l_ol + o.link_off;
  const float *lg = hp.l_g + o.link_off, *la = hp.l_a + o.link_off;
  const int32_t start = d->fst->start_state;
  const float inf = std::numeric_limits<float>::infinity();
  static thread_local std::vector<int32_t> cnt, ord, parent;
  static thread_local std::vector<LatWeight> dist;
  int max_f = 0;
  for (int k = 0; k < n; k++) {
    if (tf[k] < 0) return 1;
    max_f = std::max(max_f, tf[k]);
  }
  cnt.assign(2 * static_cast<size_t>(max_f) + 3, 0);
  for (int j = 0; j < m; j++) {
    if (ls[j] < 0 || ls[j] >= n || ld[j] < 0 || ld[j] >= n) return 1;
    const int fs = tf[ls[j]], fd = tf[ld[j]];
    if (fd != fs && fd != fs + 1) return 1;
    cnt[2 * fd + (fd == fs ? 1 : 0) + 1]++;
  }
  for (size_t b = 0; b + 1 < cnt.size(); b++) cnt[b + 1] += cnt[b];
  ord.resize(m);
  {
    static thread_local std::vector<int32_t> fill;
    fill.assign(cnt.begin(), cnt.end() - 1);
    for (int j = 0; j < m; j++) {
      const int fs = tf[ls[j]], fd = tf[ld[j]];
      ord[fill[2 * fd + (fd == fs ? 1 : 0)]++] = j;
    }
  }
  // canonical state 0: the first token of frame 0 in the order (start token first, HCLG state)
  int s0 = -1;
  for (int k = 0; k < n; k++) {
    if (tf[k] != 0) continue;
    if (s0 < 0) { s0 = k; continue; }
    const bool xs = ts[k] != start, ys = ts[s0] != start;
    if (xs != ys ? xs < ys : ts[k] < ts[s0]) s0 = k;
  }
  if (s0 < 0) return 1;
  dist.assign(n, LatWeight{inf, inf});
  parent.assign(n, -1);
  dist[s0] = LatWeight{0.f, 0.f};
  // is link j in front of link p in the canonical arc order?  (both end in the same state)
  auto before = [&](int j, int p) {
    const int a = ls[j], b = ls[p];
    if (a != b) {
      if (tf[a] != tf[b]) return tf[a] < tf[b];
      if (tf[a] == 0) {
        const bool xs = ts[a] != start, ys = ts[b] != start;
        if (xs != ys) return xs < ys;
      }
      if (ts[a] != ts[b]) return ts[a] < ts[b];
    }
    if (li[j] != li[p]) return li[j] < li[p];
    if (lo[j] != lo[p]) return lo[j] < lo[p];
    if (lg[j] != lg[p]) return lg[j] < lg[p];
    return la[j] < la[p];
  };
  auto relax = [&](int j) -> bool {
    const LatWeight sd = dist[ls[j]];
    if (sd.v1 == inf) return false;
    const LatWeight w{sd.v1 + lg[j], sd.v2 + la[j]};
    LatWeight &nd = dist[ld[j]];
    const int c = (nd.v1 == inf && nd.v2 == inf) ? 1 : Compare(w, nd);
    if (c == 1 || (c == 0 && parent[ld[j]] != j && (parent[ld[j]] < 0 || before(j, parent[ld[j]])))) {
      nd = w;
      parent[ld[j]] = j;
      return true;
    }
    return false;
  };
  for (int f = 0; f <= max_f; f++) {
    for (int q = cnt[2 * f]; q < cnt[2 * f + 1]; q++) (void)relax(ord[q]);   // emitting links into frame f
    const int eb = cnt[2 * f + 1], ee = cnt[2 * f + 2];
    for (int guard = 0; guard < ee - eb + 2; guard++) {                      // epsilon links inside frame f
      bool changed = false;
      for (int q = eb; q < ee; q++) changed |= relax(ord[q]);
      if (!changed) break;
    }
  }
  // the final state: best dist + final cost (:177-186 as BuildLattice), ties to the smaller canonical state
  const bool have_final = o.stats.reached_final != 0;
  LatWeight best{inf, inf};
  int best_state = -1;
  for (int k = 0; k < n; k++) {
    if (tf[k] != T || dist[k].v1 == inf) continue;
    float fc = 0.0f;
    if (have_final) {
      fc = d->fst->final_host[ts[k]];
      if (fc == inf) continue;
    }
    const LatWeight w{dist[k].v1 + fc, dist[k].v2 + 0.0f};
    bool take = best_state < 0;
    if (!take) {
      const int c = Compare(w, best);
      if (c == 1) take = true;
      else if (c == 0) {
        const bool xs = T == 0 && ts[k] != start, ys = T == 0 && ts[best_state] != start;
        take = xs != ys ? xs < ys : ts[k] < ts[best_state];
      }
    }
    if (take) { best = w; best_state = k; }
  }
  if (best_state < 0) {
    SetError("GetBestPath: no final state reachable for utterance %d", utt);
    return L.bp_rc = KH_ESTATE;
  }
  L.bp_ali.clear();
  L.bp_words.clear();
  for (int s = best_state; parent[s] >= 0; s = ls[parent[s]]) {
    const int j = parent[s];
    if (li[j] != 0) L.bp_ali.push_back(li[j]);
    if (lo[j] != 0) L.bp_words.push_back(lo[j]);
  }
  std::reverse(L.bp_ali.begin(), L.bp_ali.end());
  std::reverse(L.bp_words.begin(), L.bp_words.end());
  L.bp_graph = best.v1;
  L.bp_acoustic = best.v2;
  return L.bp_rc = KH_OK;
}

// Best path of utterance `utt`, cached in the Lat: from the pool (above) unless the canonical lattice exists already or
// KH_DECODER_CANONICAL_BESTPATH asks for the route over it (the cross-check of tests/test_gpu_decoder.py).
int ComputeBestPath(KhDecoder *d, int utt) {
  {
    KhDecoder::Lat &L0 = d->lats[utt];
    if (L0.bp_rc == 1 && !L0.built && !getenv("KH_DECODER_CANONICAL_BESTPATH")) {
      const int rc = ComputeBestPathLean(d, utt);
      if (rc != 1) return rc;
    }
  }
  int rc = BuildLattice(d, utt);
  if (rc) return rc;
  KhDecoder::Lat &L = d->lats[utt];
  if (L.bp_rc != 1) {
    if (L.bp_rc != KH_OK) SetError("GetBestPath: no best path for utterance %d", utt);
    return L.bp_rc;
  }
  const int ns = static_cast<int>(L.state_frame.size()), na = static_cast<int>(L.arc_src.size());
  if (ns == 0) {
    SetError("GetBestPath: empty lattice for utterance %d", utt);
    return L.bp_rc = KH_ESTATE;
  }
  const float inf = std::numeric_limits<float>::infinity();
  static thread_local std::vector<LatWeight> dist;
  static thread_local std::vector<int32_t> parent;
  dist.assign(ns, LatWeight{inf, inf});
  parent.assign(ns, -1);
  dist[0] = LatWeight{0.f, 0.f};
  bool changed = true;
  for (int guard = 0; changed && guard < ns + 2; guard++) {
    changed = false;
    for (int j = 0; j < na; j++) {
      const LatWeight sd = dist[L.arc_src[j]];
      if (sd.v1 == inf) continue;
      const LatWeight w{sd.v1 + L.arc_g[j], sd.v2 + L.arc_a[j]};
      LatWeight &nd = dist[L.arc_dst[j]];
      const int c = (nd.v1 == inf && nd.v2 == inf) ? 1 : Compare(w, nd);
      if (c == 1 || (c == 0 && parent[L.arc_dst[j]] > j)) {
        nd = w;
        parent[L.arc_dst[j]] = j;
        changed = true;
      }
    }
  }
  LatWeight best{inf, inf};
  int best_state = -1;
  for (int s = 0; s < ns; s++) {
    if (L.state_final[s] == inf || dist[s].v1 == inf) continue;
    const LatWeight w{dist[s].v1 + L.state_final[s], dist[s].v2 + 0.0f};
    if (best_state < 0 || Compare(w, best) == 1) {
      best = w;
      best_state = s;
    }
  }
  if (best_state < 0) {
    SetError("GetBestPath: no final state reachable for utterance %d", utt);
    return L.bp_rc = KH_ESTATE;
  }
  std::vector<int32_t> path;
  for (int s = best_state; parent[s] >= 0; s = L.arc_src[parent[s]]) path.push_back(parent[s]);
  std::reverse(path.begin(), path.end());
  L.bp_ali.clear();
  L.bp_words.clear();
  for (int j : path) {
    if (L.arc_il[j] != 0) L.bp_ali.push_back(L.arc_il[j]);
    if (L.arc_ol[j] != 0) L.bp_words.push_back(L.arc_ol[j]);
  }
  L.bp_graph = best.v1;
  L.bp_acoustic = best.v2;
  return L.bp_rc = KH_OK;
}

// DeterminizeLatticePhonePrunedWrapper (decoder-wrappers.cc:264-274) on the utterance's raw lattice, cached.
int DeterminizeUtt(KhDecoder *d, int utt) {
  if (d->clats[utt]) return KH_OK;
  int rc = BuildLattice(d, utt);
  if (rc) return rc;
  const KhDecoder::Lat &L = d->lats[utt];
  KhCompactLattice *c = kh_determinize_lattice_phone_pruned(
      static_cast<int>(L.state_frame.size()), static_cast<int>(L.arc_src.size()), L.arc_src.data(), L.arc_dst.data(), L.arc_il.data(),
      L.arc_ol.data(), L.arc_g.data(), L.arc_a.data(), L.state_final.data(), d->det_phone ? d->det_tid_phone.data() : nullptr,
      static_cast<int>(d->det_tid_phone.size()), d->det_beam, d->det_delta, d->det_max_mem, d->det_phone, d->det_word, d->det_minimize);
  if (!c) return KH_EINVAL;
  d->clats[utt] = c;
  return KH_OK;
}

// Arena slab of the decoder: n_want slots sized for utterances of up to T_max frames
// (fewer if they do not fit in free memory); establishes the arena invariants.
int EnsureSlots(KhDecoder *d, int n_want, int T_max, hipStream_t st, int *n_slots_out, int scale = 1) {
  int n_slots = n_want;
  const bool same_kind = d->slab_lazy == d->lazy && d->slab_link_a == d->alloc_link_a && d->slab_exact == d->exact;
  if (T_max <= d->slab_T && scale == d->slab_scale && same_kind) n_slots = std::min(n_slots, d->slot_limit);  // an earlier batch found that more do not fit
  const long long kCap = (1ll << 28);
  const int tfc = static_cast<int>(std::min<long long>(kCap, 1ll * d->tok_frame_cap * scale)),
            lfc = static_cast<int>(std::min<long long>(kCap, 1ll * d->link_frame_cap * scale)),
            wt = static_cast<int>(std::min<long long>(kCap, 1ll * d->win_tok * scale)),
            wl = static_cast<int>(std::min<long long>(kCap, 1ll * d->win_link * scale)),
            et = static_cast<int>(std::min<long long>(kCap, 1ll * d->expected_tokens * scale));
  if (n_slots > d->slab_slots || T_max > d->slab_T || scale != d->slab_scale || !same_kind) {
    PoolFree(d->slab);
    d->slab = nullptr;
    size_t slab_bytes = 0;
    // leave room for the lattice pool and the caller: all but 16 GB of what is free
    // (blocks cached by the library's own pool count as free: PoolMalloc returns
    // them to HIP when an allocation fails)
    size_t free_b = 0, total_b = 0;
    KH_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += PoolCachedBytes();
    const size_t budget = free_b > (16ull << 30) ? free_b - (16ull << 30) : free_b / 2;
    // the lazy schedule takes what memory it can use, not what it must have: it leaves 32 GB more to the caller
    size_t lazy_budget = budget > (64ull << 30) ? budget - (32ull << 30) : budget / 2;
    if (const char *e = getenv("KH_DECODER_ARENA_GB")) lazy_budget = std::min<size_t>(budget, static_cast<size_t>(atof(e) * (1ull << 30)));
    const int want_slots = n_slots;
    const ArenaCaps floor_caps = WindowedCaps(T_max, tfc, lfc, wt, wl, d->cfg.prune_interval);
    ArenaCaps caps = floor_caps;
    auto slot_bytes = [&](const ArenaCaps &c) {
      Carver sizer{nullptr};
      Utt tmp;
      CarveSlot(sizer, tmp, T_max, tfc, lfc, c, d->alloc_link_a != 0, d->cfg.hash_ratio, et, d->exact != 0, nullptr, scale);
      return sizer.off;
    };
    for (;; n_slots = (n_slots + 1) / 2) {
      caps = floor_caps;
      if (d->lazy) {
        const ArenaCaps want = LazyCaps(floor_caps, T_max, tfc, lfc, wt);
        const size_t b_floor = slot_bytes(floor_caps), b_want = slot_bytes(want), per_slot = lazy_budget / n_slots;
        if (b_want <= per_slot) {
          caps = want;
        } else if (per_slot > b_floor && b_want > b_floor) {
          // (the bytes of a slot are affine in its capacities up to alignment: interpolate, a little under)
          const double r = 0.999 * static_cast<double>(per_slot - b_floor) / static_cast<double>(b_want - b_floor);
          caps.tok = floor_caps.tok + static_cast<long long>(r * (want.tok - floor_caps.tok));
          caps.link = floor_caps.link + static_cast<long long>(r * (want.link - floor_caps.link));
        }
      }
      slab_bytes = slot_bytes(caps) * n_slots;
      if (slab_bytes <= budget || n_slots == 1) {
        d->slab = PoolMalloc(slab_bytes);
        if (d->slab || n_slots == 1) break;
      }
    }
    if (!d->slab) { d->slab_bytes = 0; d->slab_slots = 0; return KH_ENOMEM; }
    if (getenv("KH_DECODER_PROFILE"))
      fprintf(stderr, "[kh_decoder profile] arenas (scale %d, %s schedule): %d slots (wanted %d) x %.1f MB = %.1f GB (%lld tokens, %lld links per slot); device memory free %.1f GB of %.1f GB\n",
              scale, d->lazy ? "lazy" : "windowed", n_slots, want_slots, slab_bytes / 1e6 / n_slots, slab_bytes / 1e9, caps.tok, caps.link, free_b / 1e9, total_b / 1e9);
    d->slot_limit = n_slots < want_slots ? n_slots : std::numeric_limits<int>::max();
    Carver sizer{nullptr};
    sizer.off = slab_bytes;
    d->slab_bytes = sizer.off;
    d->slab_slots = n_slots;
    d->slab_T = T_max;
    d->slab_scale = scale;
    d->slab_lazy = d->lazy;
    d->slab_link_a = d->alloc_link_a;
    d->slab_exact = d->exact;
    d->h_slots.assign(n_slots, Utt());
    Carver carver{static_cast<char *>(d->slab)};
    d->h_slotsx.assign(n_slots, UttX());
    for (int i = 0; i < n_slots; i++)
      CarveSlot(carver, d->h_slots[i], T_max, tfc, lfc, caps, d->alloc_link_a != 0, d->cfg.hash_ratio, et, d->exact != 0, &d->h_slotsx[i], scale);
    for (int i = 0; i < n_slots; i++) d->h_slotsx[i].x_rec0 = Arr<const KhInt4>((GP(const KhInt4))d->fst->rec);
    // arena invariants for the first utterance of every slot (later ones are
    // restored by the kernel): token costs = +inf, hash empty, dirty flags zero
    for (int i = 0; i < n_slots; i++) {
      Utt &u = d->h_slots[i];
      hipLaunchKernelGGL(FillU32, dim3(256), dim3(256), 0, st, (uint32_t *)u.tok_cost.p, static_cast<size_t>(u.tok_cap), kEncInf);
      KH_HIP(hipMemsetAsync((void *)(unsigned long long *)u.hash.p, 0, sizeof(unsigned long long) * (static_cast<size_t>(u.hash_mask) + 1), st));
      KH_HIP(hipMemsetAsync((void *)(int32_t *)u.tmp_dirty.p, 0, sizeof(int32_t) * u.tok_frame_cap, st));
      const UttX &x = d->h_slotsx[i];
      if (x.x_hcap > 0) KH_HIP(hipMemsetAsync((void *)(uint32_t *)x.x_bmin.p, 0xFF, sizeof(uint32_t) * static_cast<size_t>(x.x_hcap), st));
    }
    PoolFree(d->d_slots);
    PoolFree(d->d_slotsx);
    d->d_slots = static_cast<Utt *>(PoolMalloc(sizeof(Utt) * n_slots));
    d->d_slotsx = static_cast<UttX *>(PoolMalloc(sizeof(UttX) * n_slots));
    if (!d->d_slots || !d->d_slotsx) return KH_ENOMEM;
    KH_HIP(hipMemcpyAsync(d->d_slotsx, d->h_slotsx.data(), sizeof(UttX) * n_slots, hipMemcpyHostToDevice, st));
  } else {
    // slots were left with dirty token costs by the previous call: refill
    for (int i = 0; i < n_slots; i++) {
      Utt &u = d->h_slots[i];
      hipLaunchKernelGGL(FillU32, dim3(256), dim3(256), 0, st, (uint32_t *)u.tok_cost.p, static_cast<size_t>(u.tok_cap), kEncInf);
    }
  }
  *n_slots_out = n_slots;
  return KH_OK;
}

// The decoder's own copy of the state records with the PDF in the first word of every arc unit
// (tid2pdf[ilabel], or ilabel - 1 without a map): the expansion needs the score column, not the
// transition-id, and reads it with the arc instead of gathering the map.  Rebuilt on every call
// (the map is the caller's and may change between calls): one pass over the label table.
// order_ids = 1 (reference order): the second word of an arc - its output label, which the search never reads - carries the
// CALLER'S id of the arc's destination state instead: what HashList hashes (state % hash_size).  The expansion reads it with
// the arc; looking it up per new token (unit id -> caller's id, a 4-byte gather into a 40 MB table per token and frame) was
// 1.2 TB of the reference-order kernel's 4.4 TB of reads.  The export reads output labels from the graph's own records then.
__global__ void ArcPdfKernel(const int32_t *__restrict__ unit_ilabel, long long n, const int32_t *__restrict__ tid2pdf, int4 *__restrict__ rec,
                             int num_cols, int *__restrict__ bad, const int4 *__restrict__ rec0, int order_ids) {
  for (long long a = blockIdx.x * static_cast<long long>(blockDim.x) + KH_TIDX; a < n; a += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int32_t il = unit_ilabel[a];
    if (il > 0) {
      rec[a].y = order_ids ? -1 - unit_ilabel[rec0[a].w & kStateMask] : rec0[a].y;
      int32_t pdf = tid2pdf ? tid2pdf[il] : il - 1;
      // the decoder gathers the score row at this column unchecked: a map entry outside the matrix is reported (first
      // offender: transition-id, pdf) and clamped so that the launch that follows cannot read out of bounds
      if (pdf < 0 || pdf >= num_cols) {
        if (atomicCAS(&bad[0], 0, 1) == 0) { bad[1] = il; bad[2] = pdf; }
        pdf = 0;
      }
      rec[a].x = pdf;
    }
  }
}

// Validates, for every caller of the C-ABI (the Python wrapper checks its host copy of the map as well), that every
// transition-id ON AN ARC of the graph maps to a column of the score matrix (the C++ mirror and direct callers
// used to reach the gathers unchecked).
int BuildArcPdf(KhDecoder *d, Params *p, const int32_t *tid2pdf, int ll_stride, hipStream_t st) {
  const size_t bytes = sizeof(int4) * static_cast<size_t>(d->fst->num_units);
  if (d->rec == nullptr) {
    d->rec = static_cast<int4 *>(PoolMalloc(bytes));
    if (d->rec == nullptr) return KH_ENOMEM;
    KH_HIP(hipMemcpyAsync(d->rec, d->fst->rec, bytes, hipMemcpyDeviceToDevice, st));
  }
  if (d->d_bad == nullptr) {
    d->d_bad = static_cast<int *>(PoolMalloc(sizeof(int) * 4));
    if (d->d_bad == nullptr) return KH_ENOMEM;
  }
  KH_HIP(hipMemsetAsync(d->d_bad, 0, sizeof(int) * 4, st));
  hipLaunchKernelGGL(ArcPdfKernel, dim3(NumCUs() * 8), dim3(256), 0, st, (const int32_t *)d->fst->unit_ilabel,
                     static_cast<long long>(d->fst->num_units), tid2pdf, d->rec,
                     ll_stride > 0 ? ll_stride : std::numeric_limits<int>::max(),   // (a launch without score rows: InitDecoding / FinalizeDecoding jobs)
                     d->d_bad, (const int4 *)d->fst->rec, d->exact ? 1 : 0);
  KH_LAUNCH_CHECK();
  int bad[4] = {0, 0, 0, 0};
  KH_HIP(hipMemcpyAsync(bad, d->d_bad, sizeof(bad), hipMemcpyDeviceToHost, st));
  KH_HIP(hipStreamSynchronize(st));
  if (bad[0] != 0) {
    SetError("the transition-id -> pdf map sends transition-id %d (on an arc of the graph) to pdf %d, outside the %d columns of the "
             "log-likelihood matrix", bad[1], bad[2], ll_stride);
    return KH_EINVAL;
  }
  d->rec_order_ids = d->exact ? 1 : 0;
  p->rec = (GP(const KhInt4))d->rec;
  return KH_OK;
}

// OFF by default (0).  With the span on, the serving stress harness saw what it never saw without it: two memory-access
// faults and one workgroup that never left its action in ~45 runs of the two serving legs (15 of 15 clean with the span off
// on the same box, 116 of 116 in round 5) - about one bad collection in several hundred thousand.  The collections
// themselves are the offline kernel's (PruneActiveTokens + full compaction), only far more frequent; the defect has not been
// found, so the switch stays an experiment: KH_SERVE_LAZY_SPAN=128 gives chunk latency max 113 -> 28 ms at 256 streams.
static int OnlineLazySpan() {
  if (const char *e = getenv("KH_SERVE_LAZY_SPAN")) return std::max(0, atoi(e));
  return 0;
}

void FillParams(const KhDecoder *d, Params *pp, int ll_stride, const int32_t *tid2pdf) {
  Params &p = *pp;
  (void)tid2pdf;
  p.rec = (GP(const KhInt4))d->rec;   // (BuildArcPdf allocates it on the first call and sets it again)
  p.n_arcs = (GP(const KhInt4))d->fst->n_arcs;
  p.unit_ilabel = (GP(const int32_t))d->fst->unit_ilabel;
  p.start = d->fst->start;
  p.num_units = static_cast<int32_t>(d->fst->num_units);
  p.num_eps = static_cast<int32_t>(d->fst->num_eps);
  p.start_has_eps = d->fst->start_has_eps;
  p.max_emit = d->fst->max_emit;
  p.lazy_span = 0;
  if (const char *e = getenv("KH_DECODER_NO_MID_SCAN")) { if (atoi(e) != 0) p.max_emit = 1 << 16; }   // (tests: the scan through memory for every frame beyond the LDS tier)
  // the frame's score row fits in LDS (two workgroups per CU share 160 KB): stage it
  // the score row shares the workgroup's 64 KB of LDS with the static block (Shared)
  p.ll_cols = (sizeof(float) * static_cast<size_t>(ll_stride) + sizeof(Shared) + 256 <= 78 * 1024) ? ll_stride : 0;
  p.keep_ac = 1;
  p.lazy_prune = 0;
  p.exact_order = 0;
  p.hash_ratio = d->cfg.hash_ratio;
  p.cl_max_load = kClMaxLoad;
  if (const char *e = getenv("KH_DECODER_CLOSURE_CAP")) p.cl_max_load = std::max(0, std::min(kClMaxLoad, atoi(e)));
  if (getenv("KH_DECODER_NO_LDS_SCORES")) p.ll_cols = 0;
  p.max_tid = d->fst->max_ilabel;
  p.beam = d->cfg.beam;
  p.lattice_beam = d->cfg.lattice_beam;
  p.beam_delta = d->cfg.beam_delta;
  p.prune_scale = d->cfg.prune_scale;
  p.max_active = d->cfg.max_active;
  p.min_active = d->cfg.min_active;
  p.prune_interval = d->cfg.prune_interval;
}

// The same pool in pinned HOST memory (offline decoding: the kernel writes finished lattices
// there and the host reads them while the kernel keeps running).
int EnsureHostPool(KhDecoder *d, long long pool_tok, long long pool_link) {
  Carver sizer{nullptr};
  sizer.Take<int32_t>(pool_tok); sizer.Take<int32_t>(pool_tok);
  for (int k = 0; k < 4; k++) sizer.Take<int32_t>(pool_link);
  sizer.Take<float>(pool_link); sizer.Take<float>(pool_link);
  if (sizer.off > d->hp_bytes) {
    if (d->hp_slab) (void)hipHostFree(d->hp_slab);
    d->hp_slab = nullptr;
    d->hp_bytes = 0;
    if (hipHostMalloc(&d->hp_slab, sizer.off, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      SetError("kh_decoder_decode: cannot pin %.1f GB of host memory for the lattice pool", sizer.off / 1e9);
      return KH_ENOMEM;
    }
    d->hp_bytes = sizer.off;
  }
  void *dev = nullptr;
  KH_HIP(hipHostGetDevicePointer(&dev, d->hp_slab, 0));
  Carver c{static_cast<char *>(dev)};
  d->hpool.t_frame = c.Take<int32_t>(pool_tok);
  d->hpool.t_state = c.Take<int32_t>(pool_tok);
  d->hpool.l_src = c.Take<int32_t>(pool_link);
  d->hpool.l_dst = c.Take<int32_t>(pool_link);
  d->hpool.l_il = c.Take<int32_t>(pool_link);
  d->hpool.l_ol = c.Take<int32_t>(pool_link);
  d->hpool.l_g = c.Take<float>(pool_link);
  d->hpool.l_a = c.Take<float>(pool_link);
  d->hpool.tok_cap = pool_tok;
  d->hpool.link_cap = pool_link;
  d->hpool.used = (GP(unsigned long long))d->d_used;
  Carver h{static_cast<char *>(d->hp_slab)};
  d->hview.t_frame = (const int32_t *)h.Take<int32_t>(pool_tok);
  d->hview.t_state = (const int32_t *)h.Take<int32_t>(pool_tok);
  d->hview.l_src = (const int32_t *)h.Take<int32_t>(pool_link);
  d->hview.l_dst = (const int32_t *)h.Take<int32_t>(pool_link);
  d->hview.l_il = (const int32_t *)h.Take<int32_t>(pool_link);
  d->hview.l_ol = (const int32_t *)h.Take<int32_t>(pool_link);
  d->hview.l_g = (const float *)h.Take<float>(pool_link);
  d->hview.l_a = (const float *)h.Take<float>(pool_link);
  return KH_OK;
}

// Lattice pool of pool_tok tokens / pool_link links (grown on demand).
int EnsurePool(KhDecoder *d, long long pool_tok, long long pool_link) {
    {
      Carver sizer{nullptr};
      sizer.Take<int32_t>(pool_tok); sizer.Take<int32_t>(pool_tok);
      for (int k = 0; k < 4; k++) sizer.Take<int32_t>(pool_link);
      sizer.Take<float>(pool_link); sizer.Take<float>(pool_link);
      if (sizer.off > d->pool_bytes) {
        PoolFree(d->pool_slab);
        d->pool_slab = PoolMalloc(sizer.off);
        if (!d->pool_slab) { d->pool_bytes = 0; return KH_ENOMEM; }
        d->pool_bytes = sizer.off;
      }
      Carver c{static_cast<char *>(d->pool_slab)};
      d->pool.t_frame = c.Take<int32_t>(pool_tok);
      d->pool.t_state = c.Take<int32_t>(pool_tok);
      d->pool.l_src = c.Take<int32_t>(pool_link);
      d->pool.l_dst = c.Take<int32_t>(pool_link);
      d->pool.l_il = c.Take<int32_t>(pool_link);
      d->pool.l_ol = c.Take<int32_t>(pool_link);
      d->pool.l_g = c.Take<float>(pool_link);
      d->pool.l_a = c.Take<float>(pool_link);
      d->pool.tok_cap = pool_tok;
      d->pool.link_cap = pool_link;
      d->pool.used = (GP(unsigned long long))d->d_used;
    }
  return KH_OK;
}

// One D2H per pool array (the lattices the reference would build on the host in
// GetRawLattice) into a new HostPool.
int FetchPool(KhDecoder *d, const unsigned long long *used, long long pool_tok, long long pool_link, hipStream_t st) {
    d->rounds.emplace_back();
    KhDecoder::HostPool &hp = d->rounds.back();
    const size_t ut = std::min<unsigned long long>(used[0], pool_tok), ul = std::min<unsigned long long>(used[1], pool_link);
    hp.t_frame.resize(ut); hp.t_state.resize(ut);
    hp.l_src.resize(ul); hp.l_dst.resize(ul); hp.l_il.resize(ul); hp.l_ol.resize(ul);
    hp.l_g.resize(ul); hp.l_a.resize(ul);
#define D2H(dst, src, n, type) if (n) KH_HIP(hipMemcpyAsync(dst.data(), src, sizeof(type) * (n), hipMemcpyDeviceToHost, st))
    D2H(hp.t_frame, (int32_t *)d->pool.t_frame, ut, int32_t);
    D2H(hp.t_state, (int32_t *)d->pool.t_state, ut, int32_t);
    D2H(hp.l_src, (int32_t *)d->pool.l_src, ul, int32_t);
    D2H(hp.l_dst, (int32_t *)d->pool.l_dst, ul, int32_t);
    D2H(hp.l_il, (int32_t *)d->pool.l_il, ul, int32_t);
    D2H(hp.l_ol, (int32_t *)d->pool.l_ol, ul, int32_t);
    D2H(hp.l_g, (float *)d->pool.l_g, ul, float);
    D2H(hp.l_a, (float *)d->pool.l_a, ul, float);
#undef D2H
    KH_HIP(hipStreamSynchronize(st));
  return KH_OK;
}

void PrintPhases(const std::vector<long long> &h_phase, int grid, int round, int np, float ms) {
  static const char *names[16] = {"cutoff", "emit_pass1", "emit_pass2", "eps_closure", "eps_links", "clear_hash",
                                  "prune", "compact", "finalize", "export", "", "", "", "", "", "other"};
  long long tot[NPH] = {0};
  long long all = 0;
  for (int i = 0; i < grid; i++)
    for (int k = 0; k < NPH; k++) {
      tot[k] += h_phase[NPH * i + k];
      if (k < 16) all += h_phase[NPH * i + k];
    }
  fprintf(stderr, "[kh_decoder profile] launch %d: %d utterances, kernel %.1f ms, %d slots; share of shader cycles:",
          round, np, ms, grid);
  all -= tot[10] + tot[11] + tot[12] + tot[13] + tot[14];
  all += tot[45];
  for (int k = 0; k < 16; k++)
    if (tot[k] && (k < 10 || k == 15)) fprintf(stderr, " %s=%.1f%%", names[k], 100.0 * tot[k] / all);
  if (tot[47]) fprintf(stderr, " [per frame: %.0f tokens, %.0f with epsilon arcs from the emitting pass, %.0f in all, %.0f epsilon link slots, %.0f tokens from the closure]",
                       double(tot[46]) / tot[47], double(tot[52]) / tot[47], double(tot[53]) / tot[47], double(tot[54]) / tot[47], double(tot[55]) / tot[47]);
  if (tot[45]) fprintf(stderr, " list_order=%.1f%% (ranks + buckets %.1f%%, queue order %.1f%%, replay %.1f%%, positions %.1f%% of it; %lld frames from LDS, %lld by the sort; "
                       "%lld of the %lld candidates under the running cutoff)", 100.0 * tot[45] / all,
                       100.0 * tot[48] / tot[45], 100.0 * tot[49] / tot[45], 100.0 * tot[50] / tot[45], 100.0 * tot[51] / tot[45], tot[56], tot[57], tot[58], tot[31]);
  if (tot[45]) fprintf(stderr, " [frames of <= 8160 / <= 16384 / more tokens: %lld / %lld / %lld; cycles per frame from LDS %.0f, by the sort %.0f; closure order by walks in %lld frames, by the queue in %lld; %lld frames with every intermediate in LDS]",
                       tot[61], tot[62], tot[63], tot[56] + tot[66] ? double(tot[59]) / (tot[56] + tot[66]) : 0.0, tot[57] ? double(tot[60]) / tot[57] : 0.0, tot[64], tot[65], tot[66]);
#ifdef KH_X_STAMPS
  if (tot[47]) {
    fprintf(stderr, "\n[kh_decoder profile] fine stamps, cycles per frame:");
    for (int k = 0; k < 64; k++) if (tot[96 + k]) fprintf(stderr, " %d:%.0f", k, double(tot[96 + k]) / tot[47]);
  }
#endif
  fprintf(stderr, "\n[kh_decoder profile] PruneActiveTokens calls %lld, frames pruned %lld (%.1f per call), tokens scanned "
          "per pruned frame %.0f, eps iterations per pruned frame %.2f, eps-closure rounds %lld\n",
          tot[12], tot[10], tot[12] ? double(tot[10]) / tot[12] : 0.0, tot[10] ? double(tot[11]) / tot[10] : 0.0,
          tot[10] ? double(tot[14]) / tot[10] : 0.0, tot[13]);
  fprintf(stderr, "[kh_decoder profile] prune by frame size: <=1024 tokens: %lld visits, %.1f%% of prune cycles (%.0f cycles each); "
          "larger: %lld visits, %.1f%% (%.0f cycles each)\n",
          tot[18], tot[6] ? 100.0 * tot[16] / tot[6] : 0.0, tot[18] ? double(tot[16]) / tot[18] : 0.0,
          tot[19], tot[6] ? 100.0 * tot[17] / tot[6] : 0.0, tot[19] ? double(tot[17]) / tot[19] : 0.0);
  fprintf(stderr, "[kh_decoder profile] PruneForwardLinks on frames > %d tokens, share of prune cycles: token init %.1f%%, "
          "emitting links %.1f%%, token + epsilon sweeps %.1f%%, excise + flags %.1f%%\n",
          NT, tot[6] ? 100.0 * tot[20] / tot[6] : 0.0, tot[6] ? 100.0 * tot[21] / tot[6] : 0.0,
          tot[6] ? 100.0 * tot[22] / tot[6] : 0.0, tot[6] ? 100.0 * tot[23] / tot[6] : 0.0);
  fprintf(stderr, "[kh_decoder profile] eps closure (thread 0), share of its cycles: list/flag/cost/state/offsets %.1f%%, arcs + wave combine %.1f%%, "
          "FindOrAdd + min + queue %.1f%%, round barrier %.1f%%; tokens processed per round %.1f; frames whose closure ran in LDS %lld, "
          "through the general routine %lld\n",
          tot[3] ? 100.0 * tot[27] / tot[3] : 0.0, tot[3] ? 100.0 * tot[28] / tot[3] : 0.0, tot[3] ? 100.0 * tot[29] / tot[3] : 0.0,
          tot[3] ? 100.0 * tot[30] / tot[3] : 0.0, tot[13] ? double(tot[33]) / tot[13] : 0.0, tot[38], tot[39]);
  fprintf(stderr, "[kh_decoder profile] emitting pass: %lld candidates materialised (%lld counted as accepted in the frames with more than 11000)\n", tot[31], tot[32]);
  fprintf(stderr, "[kh_decoder profile] compaction, share of its cycles: tokens %.1f%%, +inf fill and boundary links %.1f%%, links %.1f%%\n",
          tot[7] ? 100.0 * tot[24] / tot[7] : 0.0, tot[7] ? 100.0 * tot[25] / tot[7] : 0.0, tot[7] ? 100.0 * tot[26] / tot[7] : 0.0);
  fprintf(stderr, "[kh_decoder profile] final backward pass: %lld dense visits, %lld through the general routines, %lld hand-offs through memory; survivors %lld tokens, %lld links\n",
          tot[40], tot[41], tot[42], tot[43], tot[44]);
  fprintf(stderr, "[kh_decoder profile] link compaction: %lld single-chunk barriers, %lld group barriers, %lld slots scanned, %lld links moved\n",
          tot[34], tot[35], tot[36], tot[37]);
}

}  // namespace

extern "C" {

KhFst *kh_fst_create(int32_t num_states, int32_t start, const int64_t *arc_offsets,
                     const int32_t *ilabel, const int32_t *olabel, const float *weight,
                     const int32_t *nextstate, const float *final_cost) {
  if (EnsureDevice() != KH_OK) return nullptr;
  if (num_states <= 0 || start < 0 || start >= num_states || !arc_offsets || !ilabel || !olabel ||
      !weight || !nextstate || !final_cost) {
    SetError("kh_fst_create: bad arguments");
    return nullptr;
  }
  const int64_t na = arc_offsets[num_states];
  std::vector<uint8_t> has_eps(num_states, 0), eps_dst(num_states, 0);
  std::vector<int32_t> unit_of_state(num_states);
  int64_t units = 0, n_eps_total = 0;
  for (int32_t s = 0; s < num_states; s++) {
    unit_of_state[s] = static_cast<int32_t>(units);
    units += 1;
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      if (nextstate[a] < 0 || nextstate[a] >= num_states || ilabel[a] < 0) {
        SetError("kh_fst_create: arc %lld out of range", static_cast<long long>(a));
        return nullptr;
      }
      if (ilabel[a] == 0) {
        has_eps[s] = 1;
        eps_dst[nextstate[a]] = 1;
        n_eps_total++;
      } else {
        units++;
      }
    }
    // the kernels address the tables with 32-bit byte offsets (Arr<T>), 16 bytes per unit, and a
    // state id (= unit index) shares its word with two flag bits
    if (units >= (int64_t(1) << 28) || n_eps_total >= (int64_t(1) << 28)) {
      SetError("kh_fst_create: %lld states + emitting arcs exceed the 2^28 units a 32-bit byte offset reaches", static_cast<long long>(units));
      return nullptr;
    }
  }
  static_assert(kStateMask >= (1 << 28) - 1, "unit ids must fit the state field of an arc");
  std::vector<int4> rec(static_cast<size_t>(units)), n_arcs;
  std::vector<int32_t> unit_ilabel(static_cast<size_t>(units), 0);   // (every unit is written below)
  n_arcs.reserve(static_cast<size_t>(n_eps_total));
  int32_t max_il = 0, max_emit = 0;
  for (int32_t s = 0; s < num_states; s++) {
    const int32_t base = unit_of_state[s], eps_base = static_cast<int32_t>(n_arcs.size());
    int32_t ne = 0;
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      int wbits;
      memcpy(&wbits, &weight[a], 4);
      const int32_t ns = unit_of_state[nextstate[a]] | (has_eps[nextstate[a]] ? kHasEps : 0) | (eps_dst[nextstate[a]] ? kEpsDst : 0);
      if (ilabel[a] != 0) {
        ne++;
        rec[static_cast<size_t>(base) + ne] = make_int4(ilabel[a], olabel[a], wbits, ns);
        unit_ilabel[static_cast<size_t>(base) + ne] = ilabel[a];
        max_il = std::max(max_il, ilabel[a]);
      } else {
        n_arcs.push_back(make_int4(0, olabel[a], wbits, ns));
      }
    }
    int fbits;
    memcpy(&fbits, &final_cost[s], 4);
    rec[base] = make_int4(ne, eps_base, static_cast<int32_t>(n_arcs.size()) - eps_base, fbits);
    unit_ilabel[base] = -1 - s;
    max_emit = std::max(max_emit, ne);
  }
  KhFst *f = new KhFst();
  f->num_states = num_states;
  f->start = unit_of_state[start];
  f->start_state = start;
  f->num_arcs = na;
  f->num_units = units;
  f->num_emit = units - num_states;
  f->num_eps = static_cast<int64_t>(n_arcs.size());
  f->max_ilabel = max_il;
  f->max_emit = max_emit;
  f->final_host.assign(final_cost, final_cost + num_states);
  f->start_has_eps = has_eps[start];
  auto up = [&](void **dst, const void *src, size_t bytes) -> bool {
    *dst = PoolMalloc(bytes ? bytes : 16);
    if (!*dst) return false;
    if (bytes && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) {
      SetError("kh_fst_create: upload failed");
      return false;
    }
    return true;
  };
  bool ok = up(reinterpret_cast<void **>(&f->rec), rec.data(), sizeof(int4) * rec.size()) &&
            up(reinterpret_cast<void **>(&f->unit_ilabel), unit_ilabel.data(), sizeof(int32_t) * unit_ilabel.size()) &&
            up(reinterpret_cast<void **>(&f->n_arcs), n_arcs.data(), sizeof(int4) * n_arcs.size());
  if (!ok) {
    kh_fst_destroy(f);
    return nullptr;
  }
  return f;
}

void kh_fst_destroy(KhFst *f) {
  if (!f) return;
  PoolFree(f->rec);
  PoolFree(f->unit_ilabel);
  PoolFree(f->n_arcs);
  delete f;
}

int64_t kh_fst_num_arcs(const KhFst *f) { return f ? f->num_arcs : 0; }

// The decode kernels gather tid2pdf[ilabel] and loglikes[t, pdf] unchecked: this validates,
// once per (graph, pdf map, model), what DecodableAmNnet / TransitionModel assert per call
// (decodable-am-nnet.h:76-78 "KALDI_ASSERT(transition_id ...)", transition-model.h:312).
// tid2pdf_host: HOST copy of the map, or NULL for the identity-minus-one map.
int kh_fst_check_pdf_map(const KhFst *f, const int32_t *tid2pdf_host, int n_tid2pdf, int num_cols) {
  KH_CHECK_ARG(f && num_cols > 0);
  if (!tid2pdf_host) {
    if (f->max_ilabel > num_cols) {
      SetError("kh_fst_check_pdf_map: largest ilabel %d of the graph exceeds the %d columns of the log-likelihood matrix",
               f->max_ilabel, num_cols);
      return KH_EINVAL;
    }
    return KH_OK;
  }
  if (f->max_ilabel >= n_tid2pdf) {
    SetError("kh_fst_check_pdf_map: largest ilabel %d of the graph is outside the transition-id -> pdf map (%d entries)",
             f->max_ilabel, n_tid2pdf);
    return KH_EINVAL;
  }
  for (int t = 1; t <= f->max_ilabel; t++)
    if (tid2pdf_host[t] < 0 || tid2pdf_host[t] >= num_cols) {
      SetError("kh_fst_check_pdf_map: transition-id %d maps to pdf %d, outside the %d columns of the log-likelihood matrix",
               t, tid2pdf_host[t], num_cols);
      return KH_EINVAL;
    }
  return KH_OK;
}

void kh_decoder_config_default(KhDecoderConfig *c) {
  c->beam = 16.0f;
  c->max_active = std::numeric_limits<int32_t>::max();
  c->min_active = 200;
  c->lattice_beam = 10.0f;
  c->prune_interval = 25;
  c->beam_delta = 0.5f;
  c->hash_ratio = 2.0f;
  c->prune_scale = 0.1f;
}

KhDecoder *kh_decoder_create(const KhFst *fst, const KhDecoderConfig *cfg, int max_batch,
                             int max_frames) {
  if (EnsureDevice() != KH_OK) return nullptr;
  if (!fst || !cfg || max_batch <= 0 || max_frames <= 0) {
    SetError("kh_decoder_create: bad arguments");
    return nullptr;
  }
  // LatticeFasterDecoderConfig::Check() lattice-faster-decoder.h:89-94
  if (!(cfg->beam > 0.0 && cfg->max_active > 1 && cfg->lattice_beam > 0.0 && cfg->prune_interval > 0 &&
        cfg->beam_delta > 0.0 && cfg->hash_ratio >= 1.0 && cfg->prune_scale > 0.0 && cfg->prune_scale < 1.0)) {
    SetError("kh_decoder_create: LatticeFasterDecoderConfig::Check() failed");
    return nullptr;
  }
  KhDecoder *d = new KhDecoder();
  d->fst = fst;
  d->cfg = *cfg;
  d->max_batch = max_batch;
  d->max_frames = max_frames;
  // Per-frame caps (the temporaries of a frame): the reference has no limit; a frame that
  // exceeds them reports an overflow and its utterance is decoded again with doubled arenas.
  // max_active bounds the tokens that EXPAND, not the tokens they create: a language-model
  // state of an HCLG has hundreds of arcs, and frames of 40 k new tokens occur at
  // max-active 7000.
  long long tf = 65536;
  if (const char *e = getenv("KH_DECODER_TOKENS_PER_FRAME")) tf = atoll(e);
  d->tok_frame_cap = static_cast<int>(tf);
  // tokens a frame is sized for in the hash (hash_ratio x this many entries): what a frame typically
  // holds - the reference resizes to hash_ratio x the previous frame's count (:219-225), which
  // max_active bounds before the closure adds to it - not the per-frame capacity; the table never
  // has fewer entries than that capacity (CarveSlot), so every token of a frame finds a slot
  d->expected_tokens = cfg->max_active == std::numeric_limits<int32_t>::max()
                           ? d->tok_frame_cap
                           : static_cast<int>(std::min<long long>(d->tok_frame_cap, cfg->max_active + cfg->max_active / 2));
  if (const char *e = getenv("KH_DECODER_HASH_TOKENS")) d->expected_tokens = atoi(e);
  long long lf = 4 * tf;
  if (const char *e = getenv("KH_DECODER_LINKS_PER_FRAME")) lf = atoll(e);
  d->link_frame_cap = static_cast<int>(lf);
  // average frame the compaction window is sized for
  long long wt = cfg->max_active == std::numeric_limits<int32_t>::max()
                     ? tf : std::min<long long>(tf, std::max<long long>(4096, 2ll * cfg->max_active));
  if (const char *e = getenv("KH_DECODER_WINDOW_TOKENS_PER_FRAME")) wt = atoll(e);
  d->win_tok = static_cast<int>(wt);
  long long wl = std::min<long long>(lf, 4 * wt);
  if (const char *e = getenv("KH_DECODER_WINDOW_LINKS_PER_FRAME")) wl = atoll(e);
  d->win_link = static_cast<int>(wl);
  d->max_slots = NumCUs() * KH_WG_PER_CU;  // persistent workgroups
  if (const char *e = getenv("KH_DECODER_SLOTS")) d->max_slots = std::max(1, atoi(e));
  return d;
}

int kh_decoder_set_determinize(KhDecoder *d, int enable, double beam, float delta, int64_t max_mem, const int32_t *tid_phone,
                               int n_tid, int phone_determinize, int word_determinize, int minimize) {
  KH_CHECK_ARG(d && (!enable || beam > 0.0) && (!enable || !phone_determinize || (tid_phone && n_tid > 0)));
  d->det_enable = enable != 0;
  d->det_beam = beam;
  d->det_delta = delta;
  d->det_max_mem = max_mem;
  d->det_phone = phone_determinize != 0;
  d->det_word = word_determinize != 0;
  d->det_minimize = minimize != 0;
  d->det_tid_phone.assign(tid_phone ? tid_phone : nullptr, tid_phone ? tid_phone + n_tid : nullptr);
  return KH_OK;
}

const KhCompactLattice *kh_decoder_get_compact_lattice(KhDecoder *d, int utt) {
  if (!d || utt < 0 || utt >= d->n_utts || !d->det_enable) {
    SetError("kh_decoder_get_compact_lattice: bad arguments (or kh_decoder_set_determinize was not called)");
    return nullptr;
  }
  if (DeterminizeUtt(d, utt) != KH_OK) return nullptr;
  return d->clats[utt];
}

int kh_decoder_compact_lattice_totals(KhDecoder *d, int64_t *totals) {
  KH_CHECK_ARG(d && totals && d->det_enable);
  totals[0] = totals[1] = totals[2] = totals[3] = 0;
  for (int u = 0; u < d->n_utts; u++) {
    if (d->h_out[u].stats.status != 0) continue;
    if (DeterminizeUtt(d, u) != KH_OK) return KH_EINVAL;
    int32_t ns, na, nl, nf, comp;
    kh_compact_lattice_sizes(d->clats[u], &ns, &na, &nl, &nf, &comp);
    totals[0] += ns; totals[1] += na; totals[2] += nl + nf; totals[3] += comp ? 0 : 1;
  }
  return KH_OK;
}

void kh_decoder_destroy(KhDecoder *d) {
  if (!d) return;
  d->FreeClats();
  PoolFree(d->slab);
  PoolFree(d->pool_slab);
  PoolFree(d->d_slots);
  PoolFree(d->d_slotsx);
  PoolFree(d->d_in);
  PoolFree(d->d_out);
  PoolFree(d->d_used);
  PoolFree(d->d_phase);
  PoolFree(d->rec);
  PoolFree(d->d_bad);
  if (d->hp_slab) (void)hipHostFree(d->hp_slab);
  if (d->h_out_pinned) (void)hipHostFree(d->h_out_pinned);
  if (d->h_done) (void)hipHostFree(d->h_done);
  delete d;
}

int kh_decoder_decode(KhDecoder *d, const float *loglikes, int ll_stride,
                      const int32_t *utt_off, int n_utts, const int32_t *tid2pdf) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(d && loglikes && utt_off && n_utts > 0 && n_utts <= d->max_batch && ll_stride > 0);
  hipStream_t st = Stream();
  const bool tprof = getenv("KH_DECODER_PROFILE") != nullptr;
  auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_enter = tnow();
  d->n_utts = n_utts;
  d->lats.assign(n_utts, KhDecoder::Lat());
  d->FreeClats();
  d->clats.assign(n_utts, nullptr);
  d->h_T.resize(n_utts);
  int T_max = 0;
  long long tot_frames = 0;
  for (int i = 0; i < n_utts; i++) {
    const int T = utt_off[i + 1] - utt_off[i];
    KH_CHECK_ARG(T > 0 && T <= d->max_frames);
    d->h_T[i] = T;
    T_max = std::max(T_max, T);
    tot_frames += T;
  }
  if (!d->d_in) {
    d->d_in = static_cast<UttIn *>(PoolMalloc(sizeof(UttIn) * d->max_batch));
    d->d_out = static_cast<UttOut *>(PoolMalloc(sizeof(UttOut) * d->max_batch));
    d->d_used = static_cast<unsigned long long *>(PoolMalloc(sizeof(unsigned long long) * 4));
    if (!d->d_in || !d->d_out || !d->d_used) return KH_ENOMEM;
    if (getenv("KH_DECODER_PROFILE")) {
      d->d_phase = static_cast<long long *>(PoolMalloc(sizeof(long long) * NPH * d->max_slots));
      if (!d->d_phase) return KH_ENOMEM;
    }
  }
  if (!d->h_out_pinned) {
    if (hipHostMalloc(reinterpret_cast<void **>(&d->h_out_pinned), sizeof(UttOut) * d->max_batch, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&d->h_done), sizeof(int32_t) * d->max_batch, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      SetError("kh_decoder_decode: cannot allocate pinned host memory");
      return KH_ENOMEM;
    }
  }
  // the graph's transition-ids and pdfs must lie inside the tables they index (the kernel
  // gathers tid2pdf[ilabel] and the score row unchecked)
  if (tid2pdf == nullptr && d->fst->max_ilabel > ll_stride) {
    SetError("kh_decoder_decode: the graph's largest ilabel %d exceeds the %d columns of the log-likelihood matrix",
             d->fst->max_ilabel, ll_stride);
    return KH_EINVAL;
  }
  // ---- queue order: longest first (greedy LPT over the persistent workgroups)
  d->order.resize(n_utts);
  for (int i = 0; i < n_utts; i++) d->order[i] = i;
  std::stable_sort(d->order.begin(), d->order.end(),
                   [&](int a, int b) { return d->h_T[a] > d->h_T[b]; });
  Params p;
  FillParams(d, &p, ll_stride, tid2pdf);
  // the whole score matrix stays in place until the lattices are exported: the acoustic costs of the links
  // are recomputed there instead of stored (KH_DECODER_KEEP_AC=1: stored, as the online decoder has to)
  p.keep_ac = getenv("KH_DECODER_KEEP_AC") != nullptr && atoi(getenv("KH_DECODER_KEEP_AC")) != 0 ? 1 : 0;
  // backward pruning on demand (Params::lazy_prune); KH_DECODER_PRUNE_SCHEDULE=interval: every prune_interval frames
  {
    const char *e = getenv("KH_DECODER_PRUNE_SCHEDULE");
    p.lazy_prune = (e != nullptr && (strcmp(e, "interval") == 0 || strcmp(e, "reference") == 0)) ? 0 : 1;
  }
  d->lazy = p.lazy_prune;
  d->alloc_link_a = p.keep_ac;
  // the reference's own iteration order (kh_decoder_set_reference_order; KH_DECODER_ORDER=reference|canonical overrides)
  {
    int ex = d->exact;
    if (const char *e = getenv("KH_DECODER_ORDER")) ex = strcmp(e, "reference") == 0 ? 1 : (strcmp(e, "canonical") == 0 ? 0 : ex);
    d->exact = ex;
    p.exact_order = ex;
    // KH_DECODER_ORDER_SORT=1: list positions by the comparison-free radix sort of rounds 4 (OrderFrontierSort) in every
    // frame - the path frames beyond the LDS construction's capacity take; the tests run the suites through it
    if (ex && getenv("KH_DECODER_ORDER_SORT") != nullptr && atoi(getenv("KH_DECODER_ORDER_SORT")) != 0) p.exact_order = 2;
  }
  if ((rc = BuildArcPdf(d, &p, tid2pdf, ll_stride, Stream()))) return rc;
  if (!d->ev0) {
    KH_HIP(hipEventCreate(&d->ev0));
    KH_HIP(hipEventCreate(&d->ev1));
  }
  d->h_out.assign(n_utts, UttOut());
  d->h_round.assign(n_utts, 0);
  d->rounds.clear();
  d->last_kernel_ms = 0.f;
  d->last_host_tail_ms = 0.f;
  // lattice pool in pinned host memory: ~3 x the density the recipe's options give on the
  // structured workload (17 states / 27 arcs per frame); an utterance that does not fit reports
  // its exact size and is decoded again
  long long tok_per_frame = 48, link_per_frame = 80;
  if (const char *e = getenv("KH_DECODER_POOL_TOKENS_PER_FRAME")) {
    tok_per_frame = atoll(e);
    link_per_frame = tok_per_frame * 3 / 2;
  }
  std::vector<int> pending(d->order);
  int n_workers = static_cast<int>(std::thread::hardware_concurrency());
  // never more threads than the CPU time the container may use (cgroup quota: beyond it the kernel throttles the whole
  // group, and every thread stalls)
  n_workers = std::max(1, std::min(std::min(n_workers, std::min(64, HostCpuQuota())), n_utts));
  // one process per GPU: the ranks of a node share the container's CPUs (LOCAL_WORLD_SIZE is set by torchrun)
  if (const char *e = getenv("LOCAL_WORLD_SIZE")) {
    const int ranks = atoi(e);
    if (ranks > 1) n_workers = std::max(2, n_workers / ranks);
  }
  if (const char *e = getenv("KH_DECODER_HOST_THREADS")) n_workers = std::max(1, atoi(e));
  if (static_cast<int>(d->arenas.size()) < n_workers) d->arenas.resize(n_workers);
  for (auto &a : d->arenas) a.Reset();
  // Two things are sized from estimates, and an utterance that outgrows either is decoded
  // again in a further launch while the finished ones are kept:
  //  * the lattice pool (status 6): the utterance reports its exact size, the next pool has it;
  //  * the slot arenas (status 1-4: tokens / links per frame, window): the next launch runs
  //    with arenas twice as large (the reference has no such limits).
  // An utterance that still overflows at 16 x the default arenas is left failed (its stats
  // carry the status; the getters return KH_ECAPACITY for it) - the caller counts it like the
  // reference counts a failed Decode() (num_fail, nnet-latgen-faster.cc:170).
  long long need_tok = 0, need_link = 0;
  int scale = 1;
  const int kMaxScale = 16;
  bool pool_exact = false;
  bool hook_called = false;   // kh_decoder_set_after_launch: once per call, after the last launch
  for (int round = 0; !pending.empty(); round++) {
    if (round > 8) {
      SetError("kh_decoder_decode: %d utterances still unfinished after %d launches", static_cast<int>(pending.size()), round);
      return KH_ECAPACITY;
    }
    const int np = static_cast<int>(pending.size());
    long long frames = 0;
    int T_round = 0;
    for (int ui : pending) { frames += d->h_T[ui]; T_round = std::max(T_round, d->h_T[ui]); }
    // ---- slot arenas
    int n_slots = 0;
    rc = EnsureSlots(d, std::min(np, d->max_slots), scale == 1 ? T_max : T_round, st, &n_slots, scale);
    if (rc) return rc;
    for (int i = 0; i < n_slots; i++) d->h_slots[i].ll_stride = ll_stride;
    KH_HIP(hipMemcpyAsync(d->d_slots, d->h_slots.data(), sizeof(Utt) * n_slots, hipMemcpyHostToDevice, st));
    const long long pool_tok = !pool_exact ? frames * tok_per_frame + 65536 : need_tok + 1024;
    const long long pool_link = !pool_exact ? frames * link_per_frame + 131072 : need_link + 1024;
    rc = EnsureHostPool(d, pool_tok, pool_link);
    if (rc) return rc;
    std::vector<UttIn> h_in(np);
    for (int q = 0; q < np; q++) {
      const int ui = pending[q];
      h_in[q].ll = (GP(const float))(loglikes + static_cast<size_t>(utt_off[ui]) * ll_stride);
      h_in[q].T = d->h_T[ui];
      h_in[q].pad = 0;
    }
    KH_HIP(hipMemcpyAsync(d->d_in, h_in.data(), sizeof(UttIn) * np, hipMemcpyHostToDevice, st));
    KH_HIP(hipMemsetAsync(d->d_used, 0, sizeof(unsigned long long) * 4, st));
    if (round > 0)
      for (int i = 0; i < n_slots; i++)
        hipLaunchKernelGGL(FillU32, dim3(256), dim3(256), 0, st, (uint32_t *)d->h_slots[i].tok_cost.p,
                           static_cast<size_t>(d->h_slots[i].tok_cap), kEncInf);
    const int grid = std::min(np, n_slots);
    for (int q = 0; q < np; q++) d->h_done[q] = -1;
    void *d_out_dev = nullptr, *d_done_dev = nullptr;
    KH_HIP(hipHostGetDevicePointer(&d_out_dev, d->h_out_pinned, 0));
    KH_HIP(hipHostGetDevicePointer(&d_done_dev, d->h_done, 0));
    KH_HIP(hipEventRecord(d->ev0, st));
#define KH_LAUNCH_DECODE(LAZY, EXACT)                                                                                          \
  hipLaunchKernelGGL((DecodeKernel<LAZY, EXACT>), dim3(grid), dim3(NT), DynLdsBytes(p.ll_cols), st, d->d_slots, d->d_in,          \
                     static_cast<UttOut *>(d_out_dev), np, d->hpool, p, (GP(long long))d->d_phase, (GP(int32_t))d_done_dev,       \
                     (const UttX *)d->d_slotsx)
    if (p.exact_order) {
      if (p.lazy_prune) KH_LAUNCH_DECODE(true, true); else KH_LAUNCH_DECODE(false, true);
    } else {
      if (p.lazy_prune) KH_LAUNCH_DECODE(true, false); else KH_LAUNCH_DECODE(false, false);
    }
#undef KH_LAUNCH_DECODE
    KH_LAUNCH_CHECK();
    KH_HIP(hipEventRecord(d->ev1, st));
    // ---- host threads: canonical lattice + best path of every utterance as it completes
    // (GetRawLattice + GetBestPath, decoder-wrappers.cc:215-262), overlapped with the kernel
    std::atomic<int> next_done(0);
    std::atomic<bool> kernel_done(false);
    const int round_now = round;
    (void)round_now;
    auto worker = [&](int w) {
      for (;;) {
        const int i = next_done.fetch_add(1);
        if (i >= np) break;
        int q;
        for (;;) {
          q = __atomic_load_n(&d->h_done[i], __ATOMIC_ACQUIRE);
          if (q >= 0) break;
          if (kernel_done.load()) {
            q = __atomic_load_n(&d->h_done[i], __ATOMIC_ACQUIRE);
            break;
          }
          std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        if (q < 0) break;  // the kernel ended without this completion (launch failure)
        const UttOut qo = d->h_out_pinned[q];
        if (qo.stats.status != 0) continue;  // decoded again in a later launch, or left failed
        const int ui = pending[q];
        d->h_out[ui] = qo;
        d->h_round[ui] = -1;
        KhDecoder::Lat &L = d->lats[ui];
        KhDecoder::Arena &A = d->arenas[w];
        const size_t cn = qo.n_tok, cm = qo.n_link;
        L.state_frame.bind(static_cast<int32_t *>(A.Take(4 * cn)), cn); L.state_hclg.bind(static_cast<int32_t *>(A.Take(4 * cn)), cn);
        L.state_final.bind(static_cast<float *>(A.Take(4 * cn)), cn);
        L.arc_src.bind(static_cast<int32_t *>(A.Take(4 * cm)), cm); L.arc_dst.bind(static_cast<int32_t *>(A.Take(4 * cm)), cm);
        L.arc_il.bind(static_cast<int32_t *>(A.Take(4 * cm)), cm); L.arc_ol.bind(static_cast<int32_t *>(A.Take(4 * cm)), cm);
        L.arc_g.bind(static_cast<float *>(A.Take(4 * cm)), cm); L.arc_a.bind(static_cast<float *>(A.Take(4 * cm)), cm);
        (void)ComputeBestPath(d, ui);  // (an utterance without a best path reports it from its getter)
        if (d->det_enable) (void)DeterminizeUtt(d, ui);
      }
    };
    // (joined on every exit path; a D2H copy into pageable memory would block this thread until
    // the kernel has finished, so nothing of that kind is issued before the stream is idle)
    struct Workers {
      std::vector<std::thread> th;
      std::atomic<bool> *done;
      ~Workers() {
        done->store(true);
        for (auto &t : th) t.join();
      }
    } workers{{}, &kernel_done};
    const bool overlap = !getenv("KH_DECODER_NO_HOST_OVERLAP");
    const double t_launched = tnow();
    if (overlap)
      for (int w = 0; w < n_workers; w++) workers.th.emplace_back(worker, w);
    std::vector<UttOut> q_out(np);
    unsigned long long used[4] = {0, 0, 0, 0};
    // wait for the kernel without spinning on a core (hipStreamSynchronize busy-waits): the completion threads need the
    // CPUs - every one of them where the container's quota is 16 for one GPU, and all the more with eight ranks on a node
    hipError_t sync_err = hipSuccess;
    if (overlap) {
      while ((sync_err = hipEventQuery(d->ev1)) == hipErrorNotReady) std::this_thread::sleep_for(std::chrono::microseconds(200));
      if (sync_err == hipSuccess) sync_err = hipStreamSynchronize(st);
    } else {
      sync_err = hipStreamSynchronize(st);
    }
    const double t_synced = tnow();
    kernel_done.store(true);
    // the few words the host still wants from the device, while the stream is idle (the caller's turn below may queue a
    // whole forward pass on it: nothing of this call waits behind that)
    std::vector<long long> h_phase;
    if (sync_err == hipSuccess) {
      sync_err = hipMemcpyAsync(used, d->d_used, sizeof(used), hipMemcpyDeviceToHost, st);
      if (sync_err == hipSuccess && d->d_phase) {
        h_phase.resize(NPH * static_cast<size_t>(grid));
        sync_err = hipMemcpyAsync(h_phase.data(), d->d_phase, sizeof(long long) * NPH * grid, hipMemcpyDeviceToHost, st);
      }
      if (sync_err == hipSuccess) sync_err = hipStreamSynchronize(st);
    }
    for (int q = 0; q < np; q++) q_out[q] = d->h_out_pinned[q];
    // The caller's turn (kh_decoder_set_after_launch): the LAST decode kernel of this call has finished - no utterance is
    // waiting to be decoded again from the score matrix (an arena or pool overflow: the loop below), and with
    // Params::keep_ac == 0 the export inside the kernel was the last reader of the scores - while the completion threads
    // still build / determinize this batch's lattices.  Work the hook enqueues (the next batch's forward pass, into the
    // same score buffer if the caller likes) overlaps that host tail.
    if (sync_err == hipSuccess && d->after_launch != nullptr && !hook_called) {
      bool again = false;
      for (int q = 0; q < np && !again; q++) {
        const int s6 = q_out[q].stats.status;
        again = s6 == 6 || (s6 != 0 && scale < kMaxScale);
      }
      if (!again) {
        hook_called = true;
        d->after_launch(d->after_launch_arg);
      }
    }
    if (!overlap)
      for (int w = 0; w < n_workers; w++) workers.th.emplace_back(worker, w);
    for (auto &t : workers.th) t.join();
    workers.th.clear();
    d->last_host_tail_ms += static_cast<float>(tnow() - t_synced);
    if (tprof)
      fprintf(stderr, "[kh_decoder profile] host: %.1f ms before the launch returned, %.1f ms until the stream was idle, %.1f ms more for the host threads (%d)\n",
              t_launched - t_enter, t_synced - t_launched, tnow() - t_synced, n_workers);
    if (sync_err != hipSuccess) {
      SetError("kh_decoder_decode: %s", hipGetErrorString(sync_err));
      return KH_EDEVICE;
    }
    float ms = 0.f;
    KH_HIP(hipEventElapsedTime(&ms, d->ev0, d->ev1));
    d->last_kernel_ms += ms;
#ifdef KH_BOUNDS_CHECK
    {
      int h[8];
      KH_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_oob), sizeof(h)));
      fprintf(stderr, "[kh bounds check] violations=%d first: code=%d v=%d lo=%d hi=%d thread=%d\n", h[0], h[1], h[2], h[3], h[4], h[5]);
      int z[8] = {0};
      KH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_oob), z, sizeof(z)));
    }
#endif
#ifdef KH_BARRIER_CHECK
    {
      int h[4];
      KH_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bar_misaligned), sizeof(h)));
      fprintf(stderr, "[kh barrier check] mismatches=%d first: n=%d neighbour=%d wave=%d\n", h[0], h[1], h[2], h[3]);
      int z[4] = {0, 0, 0, 0};
      KH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bar_misaligned), z, sizeof(z)));
    }
#endif
    if (d->d_phase) PrintPhases(h_phase, grid, round, np, ms);
    std::vector<int> next;
    need_tok = need_link = 0;
    bool grow = false, pool_short = false;
    int n_failed = 0;
    static const char *what[] = {"", "token arena / tokens-per-frame cap", "link arena", "links-per-frame cap",
                                 "compaction window", "LDS token table", "lattice pool", "survivor lists",
                                 "closure replay (insertions do not match the closure's tokens: a defect, not a capacity)"};
    for (int q = 0; q < np; q++) {
      const int ui = pending[q];
      const KhDecodeStats &hs = q_out[q].stats;
      if (hs.status == 6) {  // lattice did not fit in the pool: exact size is known now
        next.push_back(ui);
        need_tok += q_out[q].n_tok;
        need_link += q_out[q].n_link;
        pool_short = true;
        continue;
      }
      if (hs.status != 0) {
        if (scale < kMaxScale) {
          next.push_back(ui);
          grow = true;
          if (getenv("KH_DECODER_PROFILE"))
            fprintf(stderr, "[kh_decoder profile] utterance %d overflowed the %s at frame %d (scale %d): decoded again with 2 x the arenas\n",
                    ui, what[std::min(std::max(hs.status, 0), 8)], hs.num_frames, scale);
          continue;
        }
        // give up on this utterance only
        n_failed++;
        SetError("kh_decoder_decode: utterance %d overflowed the %s at frame %d even with %d x the default arenas "
                 "(tokens/frame cap %d, links/frame cap %d); see KH_DECODER_TOKENS_PER_FRAME / "
                 "KH_DECODER_LINKS_PER_FRAME / KH_DECODER_STABLE_TOKENS_PER_FRAME",
                 ui, what[std::min(std::max(hs.status, 0), 8)], hs.num_frames, scale, d->tok_frame_cap, d->link_frame_cap);
      }
      if (hs.status != 0) {  // left failed: its counters and status are what the getters report
        d->h_out[ui] = q_out[q];
        d->h_round[ui] = -1;
      }
    }
    (void)used;
    // Another launch follows and will write its lattices over this one's in the pool: the canonical lattices of the
    // utterances that finished in this launch are built now (they are otherwise built on first access, from the pool).
    if (!next.empty()) {
      std::vector<int> keep;
      for (int q = 0; q < np; q++)
        if (q_out[q].stats.status == 0 && !d->lats[pending[q]].built) keep.push_back(pending[q]);
      std::atomic<int> nk(0);
      auto build = [&]() {
        for (;;) {
          const int i = nk.fetch_add(1);
          if (i >= static_cast<int>(keep.size())) break;
          (void)BuildLattice(d, keep[i]);   // (an error is reported again by the getter that asks for the lattice)
        }
      };
      std::vector<std::thread> th;
      for (int w = 1; w < std::min<int>(n_workers, static_cast<int>(keep.size())); w++) th.emplace_back(build);
      build();
      for (auto &t : th) t.join();
    }
    // the utterances that need larger arenas have no lattice size yet: estimate again
    pool_exact = pool_short && !grow;
    if (grow) scale *= 2;
    pending.swap(next);
  }
  if (d->slab_scale != 1) {  // enlarged arenas are not kept: the next batch starts from the defaults
    PoolFree(d->slab);
    d->slab = nullptr;
    d->slab_bytes = 0;
    d->slab_slots = 0;
    d->slab_T = 0;
    d->slab_scale = 1;
  }
  return KH_OK;
}

int kh_decoder_set_reference_order(KhDecoder *d, int enable) {
  KH_CHECK_ARG(d);
  d->exact = enable != 0;
  return KH_OK;
}

int kh_decoder_get_search_counters(const KhDecoder *d, int utt, int64_t *counters) {
  KH_CHECK_ARG(d && counters && utt >= -1 && utt < d->n_utts);
  counters[0] = 0;
  for (int u = std::max(utt, 0); u < (utt < 0 ? d->n_utts : utt + 1); u++) counters[0] += d->h_out[u].cand_mat;
  counters[1] = d->exact;
  return KH_OK;
}

int kh_decoder_set_after_launch(KhDecoder *d, void (*fn)(void *), void *arg) {
  KH_CHECK_ARG(d);
  d->after_launch = fn;
  d->after_launch_arg = arg;
  return KH_OK;
}

int kh_decoder_last_host_tail_ms(const KhDecoder *d, float *ms) {
  KH_CHECK_ARG(d && ms);
  *ms = d->last_host_tail_ms;
  return KH_OK;
}

int kh_decoder_last_kernel_ms(const KhDecoder *d, float *ms) {
  KH_CHECK_ARG(d && ms);
  *ms = d->last_kernel_ms;
  return KH_OK;
}

// Raw per-utterance counters of the last decode (no lattice export).
int kh_decoder_get_schedule_counters(const KhDecoder *d, int utt, int32_t *counters) {
  KH_CHECK_ARG(d && counters && utt >= 0 && utt < d->n_utts);
  for (int i = 0; i < 4; i++) counters[i] = d->h_out[utt].sched[i];
  return KH_OK;
}

int kh_decoder_get_counters(const KhDecoder *d, int utt, KhDecodeStats *stats) {
  KH_CHECK_ARG(d && stats && utt >= 0 && utt < d->n_utts);
  *stats = d->h_out[utt].stats;
  return KH_OK;
}

int kh_decoder_get_stats(const KhDecoder *dc, int utt, KhDecodeStats *stats) {
  KhDecoder *d = const_cast<KhDecoder *>(dc);
  KH_CHECK_ARG(d && stats && utt >= 0 && utt < d->n_utts);
  // the raw lattice's sizes are those of the export (the canonical lattice keeps every exported token and link)
  if (d->h_out[utt].stats.status != 0) {
    SetError("utterance %d: decoder failed (code %d: %s)", utt, d->h_out[utt].stats.status, StatusText(d->h_out[utt].stats.status));
    return KH_ECAPACITY;
  }
  *stats = d->h_out[utt].stats;
  stats->num_tokens = static_cast<int32_t>(d->h_out[utt].n_tok);
  stats->num_links = static_cast<int32_t>(d->h_out[utt].n_link);
  return KH_OK;
}

int kh_decoder_get_raw_lattice(const KhDecoder *dc, int utt, int32_t *state_frame,
                               int32_t *state_hclg, float *state_final, int32_t *arc_src,
                               int32_t *arc_dst, int32_t *arc_ilabel, int32_t *arc_olabel,
                               float *arc_graph, float *arc_acoustic) {
  KhDecoder *d = const_cast<KhDecoder *>(dc);
  KH_CHECK_ARG(d && utt >= 0 && utt < d->n_utts);
  int rc = BuildLattice(d, utt);
  if (rc) return rc;
  const KhDecoder::Lat &L = d->lats[utt];
  const size_t n = L.state_frame.size(), m = L.arc_src.size();
  if (state_frame) memcpy(state_frame, L.state_frame.data(), 4 * n);
  if (state_hclg) memcpy(state_hclg, L.state_hclg.data(), 4 * n);
  if (state_final) memcpy(state_final, L.state_final.data(), 4 * n);
  if (arc_src) memcpy(arc_src, L.arc_src.data(), 4 * m);
  if (arc_dst) memcpy(arc_dst, L.arc_dst.data(), 4 * m);
  if (arc_ilabel) memcpy(arc_ilabel, L.arc_il.data(), 4 * m);
  if (arc_olabel) memcpy(arc_olabel, L.arc_ol.data(), 4 * m);
  if (arc_graph) memcpy(arc_graph, L.arc_g.data(), 4 * m);
  if (arc_acoustic) memcpy(arc_acoustic, L.arc_a.data(), 4 * m);
  return KH_OK;
}

// GetBestPath :99-105 = fst::ShortestPath on the raw lattice +
// GetLinearSymbolSequence (decoder-wrappers.cc:232-246).  Host-side, as in the
// reference.  Tie rule (OpenFst leaves exact ties to its state numbering, which
// is arbitrary in the reference): strictly better LatticeWeight wins; on an exact
// tie the smaller canonical arc index, then the smaller final state index.
int kh_decoder_get_best_path(const KhDecoder *dc, int utt, int32_t *alignment, int cap_ali,
                             int32_t *n_ali, int32_t *words, int cap_words, int32_t *n_words,
                             float *graph_cost, float *acoustic_cost) {
  KhDecoder *d = const_cast<KhDecoder *>(dc);
  KH_CHECK_ARG(d && utt >= 0 && utt < d->n_utts && n_ali && n_words && graph_cost && acoustic_cost);
  int rc = ComputeBestPath(d, utt);
  if (rc) return rc;
  const KhDecoder::Lat &L = d->lats[utt];
  const int a = static_cast<int>(L.bp_ali.size()), w = static_cast<int>(L.bp_words.size());
  *n_ali = a;
  *n_words = w;
  if ((alignment && a > cap_ali) || (words && w > cap_words)) {
    SetError("kh_decoder_get_best_path: buffers too small (alignment %d > %d or words %d > %d)", a, cap_ali, w, cap_words);
    return KH_EINVAL;
  }
  if (alignment) memcpy(alignment, L.bp_ali.data(), sizeof(int32_t) * std::min(a, std::max(cap_ali, 0)));
  if (words) memcpy(words, L.bp_words.data(), sizeof(int32_t) * std::min(w, std::max(cap_words, 0)));
  *n_ali = a;
  *n_words = w;
  *graph_cost = L.bp_graph;
  *acoustic_cost = L.bp_acoustic;
  return KH_OK;
}

// The same for utterances [first, first + n) in one call: alignments and word sequences row-concatenated
// (ali_off / words_off: n + 1 offsets), costs per utterance.  A caller that loops over a test set in C++
// calls kh_decoder_get_best_path per utterance; through a foreign-function interface the per-call cost
// (2620 utterances x 3 calls = 1 % of the benchmark's step) is what this saves.
int kh_decoder_get_best_paths(const KhDecoder *dc, int first, int n, int32_t *alignment, int64_t cap_ali, int64_t *ali_off,
                              int32_t *words, int64_t cap_words, int64_t *words_off, float *graph_cost, float *acoustic_cost) {
  KhDecoder *d = const_cast<KhDecoder *>(dc);
  KH_CHECK_ARG(d && first >= 0 && n >= 0 && first + n <= d->n_utts && alignment && ali_off && words && words_off && graph_cost &&
               acoustic_cost);
  int64_t na = 0, nw = 0;
  for (int i = 0; i < n; i++) {
    const int rc = ComputeBestPath(d, first + i);
    if (rc) return rc;
    const KhDecoder::Lat &L = d->lats[first + i];
    const int64_t a = static_cast<int64_t>(L.bp_ali.size()), w = static_cast<int64_t>(L.bp_words.size());
    if (na + a > cap_ali || nw + w > cap_words) {
      SetError("kh_decoder_get_best_paths: buffers too small at utterance %d", first + i);
      return KH_EINVAL;
    }
    ali_off[i] = na;
    words_off[i] = nw;
    if (a) memcpy(alignment + na, L.bp_ali.data(), sizeof(int32_t) * a);
    if (w) memcpy(words + nw, L.bp_words.data(), sizeof(int32_t) * w);
    na += a;
    nw += w;
    graph_cost[i] = L.bp_graph;
    acoustic_cost[i] = L.bp_acoustic;
  }
  ali_off[n] = na;
  words_off[n] = nw;
  return KH_OK;
}

// kh_decoder_get_counters and kh_decoder_get_stats of utterances [first, first + n) (either array may be NULL)
int kh_decoder_get_stats_batch(const KhDecoder *dc, int first, int n, KhDecodeStats *counters, KhDecodeStats *stats) {
  KhDecoder *d = const_cast<KhDecoder *>(dc);
  KH_CHECK_ARG(d && first >= 0 && n >= 0 && first + n <= d->n_utts);
  for (int i = 0; i < n; i++) {
    int rc = KH_OK;
    if (counters && (rc = kh_decoder_get_counters(d, first + i, &counters[i]))) return rc;
    if (stats && (rc = kh_decoder_get_stats(d, first + i, &stats[i]))) return rc;
  }
  return KH_OK;
}

// Host post-pass of a batch in parallel: GetRawLattice + GetBestPath of every
// utterance (what DecodeUtteranceLatticeFaster does per utterance after Decode(),
// decoder-wrappers.cc:215-262), on num_threads host threads (<= 0: all cores).
int kh_decoder_prepare(KhDecoder *d, int num_threads) {
  KH_CHECK_ARG(d);
  const int n = d->n_utts;
  if (n <= 0) return KH_OK;
  // (0: the CPUs the container may use - never the 256 visible cores of a box whose cgroup grants 16: beyond the quota the
  // kernel throttles the whole group, the thread that launches the next forward pass included - shared by the node's ranks)
  int nt = num_threads;
  if (nt <= 0) {
    nt = std::min(64, HostCpuQuota());
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) {
      const int ranks = atoi(e);
      if (ranks > 1) nt = std::max(2, nt / ranks);
    }
    if (const char *e = getenv("KH_DECODER_HOST_THREADS")) nt = std::max(1, atoi(e));
  }
  nt = std::max(1, std::min(std::min(nt, 128), n));
  // every lattice not built yet gets its slice of the batch store
  {
    size_t tot_n = 0, tot_m = 0;
    for (int ui = 0; ui < n; ui++)
      if (!d->lats[ui].built && d->h_out[ui].stats.status == 0) { tot_n += d->h_out[ui].n_tok; tot_m += d->h_out[ui].n_link; }
    for (int k = 0; k < 2; k++) if (d->st_i[k].size() < tot_n) d->st_i[k].resize(tot_n);
    for (int k = 2; k < 6; k++) if (d->st_i[k].size() < tot_m) d->st_i[k].resize(tot_m);
    if (d->st_f[0].size() < tot_n) d->st_f[0].resize(tot_n);
    for (int k = 1; k < 3; k++) if (d->st_f[k].size() < tot_m) d->st_f[k].resize(tot_m);
    size_t on = 0, om = 0;
    for (int ui = 0; ui < n; ui++) {
      KhDecoder::Lat &L = d->lats[ui];
      if (L.built || d->h_out[ui].stats.status != 0) continue;
      const size_t cn = d->h_out[ui].n_tok, cm = d->h_out[ui].n_link;
      L.state_frame.bind(d->st_i[0].data() + on, cn); L.state_hclg.bind(d->st_i[1].data() + on, cn);
      L.state_final.bind(d->st_f[0].data() + on, cn);
      L.arc_src.bind(d->st_i[2].data() + om, cm); L.arc_dst.bind(d->st_i[3].data() + om, cm);
      L.arc_il.bind(d->st_i[4].data() + om, cm); L.arc_ol.bind(d->st_i[5].data() + om, cm);
      L.arc_g.bind(d->st_f[1].data() + om, cm); L.arc_a.bind(d->st_f[2].data() + om, cm);
      on += cn;
      om += cm;
    }
  }
  std::atomic<int> next(0), first_rc(0);
  std::mutex mu;
  std::string first_err;
  auto work = [&]() {
    for (;;) {
      const int ui = next.fetch_add(1);
      if (ui >= n) break;
      if (d->h_out[ui].stats.status != 0) continue;  // an utterance that failed alone (capacity): its getters report it
      const int rc = ComputeBestPath(d, ui);
      if (rc != KH_OK) {
        std::lock_guard<std::mutex> l(mu);
        if (first_rc.load() == 0) { first_rc = rc; first_err = LastError(); }
      }
    }
  };
  std::vector<std::thread> th;
  for (int i = 1; i < nt; i++) th.emplace_back(work);
  work();
  for (auto &t : th) t.join();
  if (first_rc.load() != 0) {
    SetError("%s", first_err.c_str());
    return first_rc.load();
  }
  return KH_OK;
}


// ================================================================ online decoding
KhOnlineDecoder *kh_online_decoder_create(const KhFst *fst, const KhDecoderConfig *cfg, int num_streams,
                                          int max_frames) {
  KhDecoder *b = kh_decoder_create(fst, cfg, num_streams, max_frames);
  if (!b) return nullptr;
  if (const char *e = getenv("KH_DECODER_ORDER"))   // (as kh_decoder_decode reads it)
    b->exact = strcmp(e, "reference") == 0 ? 1 : (strcmp(e, "canonical") == 0 ? 0 : b->exact);
  KhOnlineDecoder *o = new KhOnlineDecoder();
  o->base = b;
  o->num_streams = num_streams;
  o->max_frames = max_frames;
  hipStream_t st = Stream();
  int n_slots = 0;
  int rc = EnsureSlots(b, num_streams, max_frames, st, &n_slots);
  if (rc == KH_OK && n_slots < num_streams) {
    SetError("kh_online_decoder_create: only %d of %d streams fit in device memory", n_slots, num_streams);
    rc = KH_ENOMEM;
  }
  if (rc == KH_OK) {
    o->d_states = static_cast<SlotState *>(PoolMalloc(sizeof(SlotState) * num_streams));
    o->d_jobs = static_cast<Job *>(PoolMalloc(sizeof(Job) * num_streams));
    b->d_out = static_cast<UttOut *>(PoolMalloc(sizeof(UttOut) * num_streams));
    b->d_used = static_cast<unsigned long long *>(PoolMalloc(sizeof(unsigned long long) * 4));
    if (!o->d_states || !o->d_jobs || !b->d_out || !b->d_used) rc = KH_ENOMEM;
  }
  if (rc == KH_OK && hipMemsetAsync(o->d_states, 0, sizeof(SlotState) * num_streams, st) != hipSuccess) rc = KH_EDEVICE;
  if (rc == KH_OK && hipMemcpyAsync(b->d_slots, b->h_slots.data(), sizeof(Utt) * num_streams, hipMemcpyHostToDevice, st) != hipSuccess) rc = KH_EDEVICE;
  if (rc == KH_OK && hipStreamSynchronize(st) != hipSuccess) rc = KH_EDEVICE;
  if (rc != KH_OK) {
    kh_online_decoder_destroy(o);
    return nullptr;
  }
  b->n_utts = num_streams;
  b->lats.assign(num_streams, KhDecoder::Lat());
  b->h_out.assign(num_streams, UttOut());
  b->h_T.assign(num_streams, 0);
  b->h_round.assign(num_streams, 0);
  o->frames.assign(num_streams, 0);
  o->inited.assign(num_streams, 0);
  o->finalized.assign(num_streams, 0);
  o->lat_key.assign(num_streams, -1);
  return o;
}

void kh_online_decoder_destroy(KhOnlineDecoder *o) {
  if (!o) return;
  (void)kh_online_decoder_serve_stop(o);
  if (o->serve_ctl) (void)hipHostFree(o->serve_ctl);
  if (o->serve_quit) (void)hipHostFree(o->serve_quit);
  if (o->serve_stream) (void)hipStreamDestroy(o->serve_stream);
  if (o->d_serve_act) PoolFree(o->d_serve_act);
  PoolFree(o->d_states);
  PoolFree(o->d_jobs);
  kh_decoder_destroy(o->base);
  delete o;
}

static int LaunchJobs(KhOnlineDecoder *o, const std::vector<Job> &jobs, int ll_stride, const int32_t *tid2pdf) {
  KhDecoder *b = o->base;
  hipStream_t st = Stream();
  Params p;
  FillParams(b, &p, ll_stride > 0 ? ll_stride : 1 << 30, tid2pdf);
  p.lazy_prune = b->lazy;   // kh_online_decoder_set_lazy_prune
  p.lazy_span = b->lazy ? OnlineLazySpan() : 0;
  if (ll_stride <= 0) p.ll_cols = 0;
  if (b->rec != nullptr && b->rec_order_ids == (b->exact ? 1 : 0) &&
      (ll_stride <= 0 || (o->pinned && tid2pdf == o->pinned_map && ll_stride == o->pinned_cols))) {
    // the records hold this map's pdfs already (kh_online_decoder_set_pdf_map) - or the jobs read no scores at all
    // (InitDecoding, FinalizeDecoding, export: only the graph's structure), and the records must not be rewritten under
    // a serving kernel that is using them
    p.rec = (GP(const KhInt4))b->rec;
  } else {
    const int rc = BuildArcPdf(b, &p, tid2pdf, ll_stride, st);
    if (rc) return rc;
    o->pinned = false;
  }
  KH_HIP(hipMemcpyAsync(o->d_jobs, jobs.data(), sizeof(Job) * jobs.size(), hipMemcpyHostToDevice, st));
  p.exact_order = b->exact ? 1 : 0;
  if (b->exact) {
    if (getenv("KH_DECODER_ORDER_SORT") != nullptr && atoi(getenv("KH_DECODER_ORDER_SORT")) != 0) p.exact_order = 2;
    hipLaunchKernelGGL(OnlineKernel<true>, dim3(static_cast<unsigned>(jobs.size())), dim3(NT), DynLdsBytes(p.ll_cols), st,
                       b->d_slots, o->d_states, o->d_jobs, b->d_out, b->pool, p, (const UttX *)b->d_slotsx);
  } else {
    hipLaunchKernelGGL(OnlineKernel<false>, dim3(static_cast<unsigned>(jobs.size())), dim3(NT), DynLdsBytes(p.ll_cols), st,
                       b->d_slots, o->d_states, o->d_jobs, b->d_out, b->pool, p, (const UttX *)nullptr);
  }
  KH_LAUNCH_CHECK();
  return KH_OK;
}

// a stream whose kernel status is not 0 -> the error the caller sees (KH_ECAPACITY: the message says which kind)
static int StreamFailed(const char *what, int stream, int code, int frame) {
  if (code == 10)
    SetError("%s: stream %d failed at frame %d (code 10: %s); the stream's utterance is lost, the other streams are not affected",
             what, stream, frame, StatusText(10));
  else
    SetError("%s: stream %d overflowed a decoder arena (code %d) at frame %d; see KH_DECODER_TOKENS_PER_FRAME / "
             "KH_DECODER_LINKS_PER_FRAME / KH_DECODER_STABLE_TOKENS_PER_FRAME", what, stream, code, frame);
  return KH_ECAPACITY;
}

// kernel status of the listed streams -> error
static int CheckStreams(KhOnlineDecoder *o, const int32_t *streams, int n, const char *what) {
  hipStream_t st = Stream();
  std::vector<SlotState> hs(o->num_streams);
  KH_HIP(hipMemcpyAsync(hs.data(), o->d_states, sizeof(SlotState) * o->num_streams, hipMemcpyDeviceToHost, st));
  KH_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < n; i++) {
    const SlotState &S = hs[streams[i]];
    if (!S.ok) return StreamFailed(what, streams[i], S.status, S.t);
    o->frames[streams[i]] = S.t;
  }
  return KH_OK;
}

static bool DistinctStreams(const KhOnlineDecoder *o, const int32_t *streams, int n) {
  std::vector<char> seen(o->num_streams, 0);
  for (int i = 0; i < n; i++) {
    if (streams[i] < 0 || streams[i] >= o->num_streams || seen[streams[i]]) return false;
    seen[streams[i]] = 1;
  }
  return true;
}

// InitDecoding (lattice-faster-online-decoder.cc:55-72) of the listed streams.
int kh_online_decoder_init_decoding(KhOnlineDecoder *o, const int32_t *streams, int n) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(o && streams && n > 0 && n <= o->num_streams && DistinctStreams(o, streams, n));
  if (o->serve_launched) {
    SetError("%s: the serving kernel owns the streams (kh_online_decoder_serve_* / kh_online_decoder_serve_stop)", "kh_online_decoder_init_decoding");
    return KH_ESTATE;
  }
  std::vector<Job> jobs(n);
  for (int i = 0; i < n; i++) {
    jobs[i].slot = streams[i];
    jobs[i].op = kJobInit;
    jobs[i].ll = (GP(const float))nullptr;
    jobs[i].ll_stride = 0;
    jobs[i].n_frames = 0;
    o->inited[streams[i]] = 1;
    o->finalized[streams[i]] = 0;
    o->lat_key[streams[i]] = -1;
  }
  rc = LaunchJobs(o, jobs, 0, nullptr);
  if (rc) return rc;
  return CheckStreams(o, streams, n, "kh_online_decoder_init_decoding");
}

// AdvanceDecoding (:747-769): stream streams[i] decodes the next num_frames[i] frames,
// whose scaled log-likelihoods are the rows of loglikes[i] (device, row stride
// ll_stride, full rows allocated).
int kh_online_decoder_advance(KhOnlineDecoder *o, const int32_t *streams, int n, const float *const *loglikes,
                              int ll_stride, const int32_t *num_frames, const int32_t *tid2pdf) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(o && streams && loglikes && num_frames && n > 0 && n <= o->num_streams && ll_stride > 0 &&
               DistinctStreams(o, streams, n));
  if (o->serve_launched) {
    SetError("kh_online_decoder_advance: the serving kernel owns the streams (kh_online_decoder_serve_publish / kh_online_decoder_serve_stop)");
    return KH_ESTATE;
  }
  std::vector<Job> jobs;
  for (int i = 0; i < n; i++) {
    const int sidx = streams[i];
    if (!o->inited[sidx] || o->finalized[sidx]) {
      SetError("kh_online_decoder_advance: stream %d: call InitDecoding() first, and not after FinalizeDecoding() "
               "(lattice-faster-online-decoder.cc:749-750)", sidx);
      return KH_ESTATE;
    }
    KH_CHECK_ARG(num_frames[i] >= 0 && o->frames[sidx] + num_frames[i] <= o->max_frames);
    if (num_frames[i] == 0) continue;
    KH_CHECK_ARG(loglikes[i] != nullptr);
    Job j;
    j.slot = sidx;
    j.op = kJobAdvance;
    // the kernel addresses the matrix by absolute frame
    j.ll = (GP(const float))(loglikes[i] - static_cast<ptrdiff_t>(o->frames[sidx]) * ll_stride);
    j.ll_stride = ll_stride;
    j.n_frames = num_frames[i];
    jobs.push_back(j);
    o->lat_key[sidx] = -1;
  }
  if (jobs.empty()) return KH_OK;
  rc = LaunchJobs(o, jobs, ll_stride, tid2pdf);
  if (rc) return rc;
  return CheckStreams(o, streams, n, "kh_online_decoder_advance");
}

int kh_online_decoder_set_pdf_map(KhOnlineDecoder *o, const int32_t *tid2pdf, int num_cols) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(o && num_cols > 0);
  Params p;
  FillParams(o->base, &p, num_cols, tid2pdf);
  if ((rc = BuildArcPdf(o->base, &p, tid2pdf, num_cols, Stream()))) return rc;
  o->pinned_map = tid2pdf;
  o->pinned_cols = num_cols;
  o->pinned = true;
  return KH_OK;
}

int kh_online_decoder_num_frames_decoded(const KhOnlineDecoder *o, int stream, int32_t *num_frames) {
  KH_CHECK_ARG(o && num_frames && stream >= 0 && stream < o->num_streams);
  *num_frames = o->frames[stream];
  return KH_OK;
}

// FinalizeDecoding (:775-790) of the listed streams.
int kh_online_decoder_finalize(KhOnlineDecoder *o, const int32_t *streams, int n) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(o && streams && n > 0 && n <= o->num_streams && DistinctStreams(o, streams, n));
  if (o->serve_launched) {
    SetError("%s: the serving kernel owns the streams (kh_online_decoder_serve_* / kh_online_decoder_serve_stop)", "kh_online_decoder_finalize");
    return KH_ESTATE;
  }
  std::vector<Job> jobs(n);
  for (int i = 0; i < n; i++) {
    const int sidx = streams[i];
    if (!o->inited[sidx] || o->finalized[sidx]) {
      SetError("kh_online_decoder_finalize: stream %d is not in a decoding run", sidx);
      return KH_ESTATE;
    }
    jobs[i].slot = sidx;
    jobs[i].op = kJobFinalize;
    jobs[i].ll = (GP(const float))nullptr;
    jobs[i].ll_stride = 0;
    jobs[i].n_frames = 0;
    o->finalized[sidx] = 1;
    o->lat_key[sidx] = -1;
  }
  rc = LaunchJobs(o, jobs, 0, nullptr);
  if (rc) return rc;
  return CheckStreams(o, streams, n, "kh_online_decoder_finalize");
}

// The offline kernel's LAZY pruning schedule for the streams (default off = the reference's: PruneActiveTokens every
// prune_interval frames): nothing is pruned while a stream advances unless its arenas run low, FinalizeDecoding prunes
// every frame once.  The final lattice, every best path and the endpointing quantities are those of the interval schedule
// (rule P, DESIGN.md section 2: the result does not depend on when the sweeps run); a raw lattice asked for BEFORE
// FinalizeDecoding is pruned as of the current frame instead of the last multiple of prune_interval.  The arenas are
// re-carved (every stream must be idle: before InitDecoding or after FinalizeDecoding + the last getter).
// The streams decode in the reference's own iteration order (kh_decoder_set_reference_order for the online decoder:
// LatticeFasterOnlineDecoder::ProcessEmitting, lattice-faster-online-decoder.cc:864-951, walks the same HashList against the
// same running next_cutoff).  Between utterances only, with the serving kernel stopped; a serving kernel started afterwards
// decodes in the same order (ServeKernel<true>).
int kh_online_decoder_set_reference_order(KhOnlineDecoder *o, int enable) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(o);
  KhDecoder *b = o->base;
  if ((b->exact != 0) == (enable != 0)) return KH_OK;
  if (o->serve_launched) {
    SetError("kh_online_decoder_set_reference_order: stop the serving kernel first");
    return KH_ESTATE;
  }
  for (int s = 0; s < o->num_streams; s++)
    if (o->inited[s] && !o->finalized[s]) {
      SetError("kh_online_decoder_set_reference_order: stream %d is in a decoding run", s);
      return KH_ESTATE;
    }
  b->exact = enable != 0;
  o->pinned = false;   // (the decoder's arc records carry state ids in this mode: built again by the next launch)
  hipStream_t st = Stream();
  int n_slots = 0;
  rc = EnsureSlots(b, o->num_streams, o->max_frames, st, &n_slots);
  if (rc == KH_OK && n_slots < o->num_streams) {
    SetError("kh_online_decoder_set_reference_order: only %d of %d streams fit in device memory", n_slots, o->num_streams);
    rc = KH_ENOMEM;
  }
  if (rc) return rc;
  KH_HIP(hipMemsetAsync(o->d_states, 0, sizeof(SlotState) * o->num_streams, st));
  KH_HIP(hipMemcpyAsync(b->d_slots, b->h_slots.data(), sizeof(Utt) * o->num_streams, hipMemcpyHostToDevice, st));
  KH_HIP(hipStreamSynchronize(st));
  for (int s = 0; s < o->num_streams; s++) {
    o->inited[s] = 0;
    o->finalized[s] = 0;
    o->frames[s] = 0;
    o->lat_key[s] = -1;
  }
  return KH_OK;
}

int kh_online_decoder_set_lazy_prune(KhOnlineDecoder *o, int enable) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(o);
  KhDecoder *b = o->base;
  if ((b->lazy != 0) == (enable != 0)) return KH_OK;
  if (o->serve_launched) {
    SetError("kh_online_decoder_set_lazy_prune: stop the serving kernel first");
    return KH_ESTATE;
  }
  for (int s = 0; s < o->num_streams; s++)
    if (o->inited[s] && !o->finalized[s]) {
      SetError("kh_online_decoder_set_lazy_prune: stream %d is in a decoding run", s);
      return KH_ESTATE;
    }
  b->lazy = enable != 0;
  hipStream_t st = Stream();
  int n_slots = 0;
  rc = EnsureSlots(b, o->num_streams, o->max_frames, st, &n_slots);
  if (rc == KH_OK && n_slots < o->num_streams) {
    SetError("kh_online_decoder_set_lazy_prune: only %d of %d streams fit in device memory", n_slots, o->num_streams);
    rc = KH_ENOMEM;
  }
  if (rc) return rc;
  KH_HIP(hipMemsetAsync(o->d_states, 0, sizeof(SlotState) * o->num_streams, st));
  KH_HIP(hipMemcpyAsync(b->d_slots, b->h_slots.data(), sizeof(Utt) * o->num_streams, hipMemcpyHostToDevice, st));
  KH_HIP(hipStreamSynchronize(st));
  for (int s = 0; s < o->num_streams; s++) { o->inited[s] = 0; o->finalized[s] = 0; o->frames[s] = 0; o->lat_key[s] = -1; }
  return KH_OK;
}

// ---- the persistent serving kernel (ServeKernel): start / stop, commands, progress
// No call below blocks for ever: every wait for the device has a deadline (KH_SERVE_TIMEOUT_MS, default 30 s) and reports
// KH_ETIMEOUT with the state of the streams' control blocks (what each workgroup was doing when it last spoke).
static double ServeTimeoutMs() {
  if (const char *e = getenv("KH_SERVE_TIMEOUT_MS")) return std::max(1.0, atof(e));
  return 30000.0;
}
#ifdef KH_SERVE_MARKERS
static int32_t *g_wave_mark_host = nullptr;
#endif
static std::string ServeDump(const KhOnlineDecoder *o, const int32_t *streams, int n) {
  std::string out;
  char buf[256];
  int shown = 0;
  // two passes: the streams whose workgroup is still there or that have something pending first (what a time-out is about:
  // round 6 had a dump whose eight entries were all workgroups that had left), then the others
  int n_alive = 0, n_pending = 0;
  for (int s = 0; s < o->num_streams; s++) {
    const ServeCtl &c = o->serve_ctl[s];
    n_alive += __atomic_load_n(&c.alive, __ATOMIC_ACQUIRE) != 0 ? 1 : 0;
    n_pending += __atomic_load_n(&c.ack_seq, __ATOMIC_ACQUIRE) != o->serve_seq[s] ? 1 : 0;
  }
  snprintf(buf, sizeof buf, " %d of %d workgroups resident, %d streams with a command in flight;", n_alive, o->num_streams, n_pending);
  out += buf;
  for (int pass = 0; pass < 2; pass++)
  for (int i = 0; i < (streams ? n : o->num_streams) && shown < 8; i++) {
    const int s = streams ? streams[i] : i;
    const ServeCtl &c = o->serve_ctl[s];
    const int ack = __atomic_load_n(&c.ack_seq, __ATOMIC_ACQUIRE), dec = __atomic_load_n(&c.decoded, __ATOMIC_ACQUIRE);
    const bool pending = ack != o->serve_seq[s] || (!o->finalized[s] && __atomic_load_n(&c.avail, __ATOMIC_ACQUIRE) > dec);
    const bool hot = pending || __atomic_load_n(&c.alive, __ATOMIC_ACQUIRE) != 0;
    if (hot != (pass == 0)) continue;
    if (streams == nullptr && !pending && __atomic_load_n(&c.hb_phase, __ATOMIC_ACQUIRE) == 0) continue;
    snprintf(buf, sizeof buf, "%s stream %d: avail %d decoded %d cmd %d/%d ack %d alive %d phase %d (to frame %d) actions %d clock %u mark %d/%d", shown ? ";" : "",
             s, c.avail, dec, c.cmd_op, o->serve_seq[s], ack, c.alive, c.hb_phase, c.hb_arg, c.hb_actions, static_cast<unsigned>(c.hb_clock),
             static_cast<int>(static_cast<uint32_t>(c.pad1[0]) >> 24), c.pad1[0] & 0xffffff);
    out += buf;
#ifdef KH_SERVE_MARKERS
    if (pass == 0 && g_wave_mark_host != nullptr && s < 1024) {
      out += " waves";
      for (int w = 0; w < 16; w++) {
        const int32_t m = __atomic_load_n(&g_wave_mark_host[64 * s + w], __ATOMIC_ACQUIRE);
        snprintf(buf, sizeof buf, " %d/%d", static_cast<int>(static_cast<uint32_t>(m) >> 24), m & 0xffffff);
        out += buf;
      }
      out += " dbg";
      for (int w = 16; w < 52; w++) {
        snprintf(buf, sizeof buf, " %d", __atomic_load_n(&g_wave_mark_host[64 * s + w], __ATOMIC_ACQUIRE));
        out += buf;
      }
    }
#endif
    shown++;
  }
  snprintf(buf, sizeof buf, "%s quit %d, launched %d, relaunches %lld", shown ? ";" : "", o->serve_quit ? *o->serve_quit : -1,
           o->serve_launched ? 1 : 0, o->serve_relaunches);
  out += buf;
  return out;
}
// waits (bounded) for the serving kernel to have ended; KH_ETIMEOUT if it has not
static int ServeJoin(KhOnlineDecoder *o, const char *what) {
  const auto t0 = std::chrono::steady_clock::now();
  const double limit = ServeTimeoutMs();
  for (;;) {
    const hipError_t q = hipStreamQuery(o->serve_stream);
    if (q == hipSuccess) break;
    (void)hipGetLastError();
    if (q != hipErrorNotReady) {
      o->serve_launched = false;
      SetError("%s: serving kernel: %s", what, hipGetErrorString(q));
      return KH_EDEVICE;
    }
    if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > limit) {
      SetError("%s: the serving kernel did not leave within %.0f ms [%s]", what, limit, ServeDump(o, nullptr, 0).c_str());
      return KH_ETIMEOUT;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  o->serve_launched = false;
  return KH_OK;
}

static int ServeEnsureRunning(KhOnlineDecoder *o) {
  if (!o->serve_ctl) {
    SetError("the serving kernel has not been started (kh_online_decoder_serve_start)");
    return KH_ESTATE;
  }
  if (o->serve_launched) {
    // running, and nobody has told it to leave: every stream has its workgroup (they leave only together)
    if (__atomic_load_n(o->serve_quit, __ATOMIC_ACQUIRE) == 0) {
      const hipError_t q = hipStreamQuery(o->serve_stream);
      if (q == hipErrorNotReady) return KH_OK;
      (void)hipGetLastError();
      if (q != hipSuccess) {
        o->serve_launched = false;
        SetError("serving kernel: %s", hipGetErrorString(q));
        return KH_EDEVICE;
      }
      o->serve_launched = false;   // (ended without being told to: cannot happen; launched again below)
    } else {
      // the grid has decided to leave (idle time): it is on its way out - launched again once it has gone
      const int rc = ServeJoin(o, "kh_online_decoder_serve");
      if (rc) return rc;
    }
    o->serve_relaunches++;
  }
  KhDecoder *b = o->base;
  hipStream_t st = Stream();
  Params p;
  FillParams(b, &p, o->serve_stride, o->serve_map);
  p.lazy_prune = b->lazy;
  p.lazy_span = b->lazy ? OnlineLazySpan() : 0;
  if (!(o->pinned && o->serve_map == o->pinned_map && o->serve_stride == o->pinned_cols && b->rec != nullptr &&
        b->rec_order_ids == (b->exact ? 1 : 0))) {
    const int rc = kh_online_decoder_set_pdf_map(o, o->serve_map, o->serve_stride);
    if (rc) return rc;
  }
  p.rec = (GP(const KhInt4))b->rec;
  KH_HIP(hipStreamSynchronize(st));   // whatever launch-per-job work is queued has written its SlotStates
  // no serving kernel is running here: what the PREVIOUS grid left in the residency fields (alive 0, phase 9 = "has left")
  // must not be read as the new grid's before its workgroups have started - kh_online_decoder_serve_poll would take the
  // freshly launched grid for one that lost a workgroup, tell it to quit and launch it a second time
  for (int s = 0; s < o->num_streams; s++) {
    __atomic_store_n(&o->serve_ctl[s].alive, 0, __ATOMIC_RELAXED);
    __atomic_store_n(&o->serve_ctl[s].hb_phase, 0, __ATOMIC_RELAXED);
  }
  __atomic_store_n(o->serve_quit, 0, __ATOMIC_RELEASE);
  void *ctl_dev = nullptr, *quit_dev = nullptr;
  KH_HIP(hipHostGetDevicePointer(&ctl_dev, o->serve_ctl, 0));
  KH_HIP(hipHostGetDevicePointer(&quit_dev, o->serve_quit, 0));
  long long idle_ticks = 200000000ll;   // 2 s of the 100 MHz wall clock
  if (const char *e = getenv("KH_SERVE_IDLE_MS")) idle_ticks = std::max(1ll, static_cast<long long>(atof(e) * 1e5));
  p.exact_order = b->exact ? 1 : 0;
  // One resident workgroup per CU while the streams fit that way.  A CU takes two of these workgroups (64 VGPRs, 73 KB of
  // LDS each) and nothing tells the dispatcher to spread 256 of them over 256 CUs: when it doubled up on ONE CU (the CUs
  // it finds busy at launch time - a fill kernel, a forward pass - get none, others get two), those two streams ran at
  // half speed for the kernel's lifetime and, the host waiting for every stream per chunk, so did the service: round 5's
  // two regimes (350 k / 200 k frames/s, chunk p50 4.7 / 9 ms, chosen per launch).  LDS is the lever: a workgroup that
  // asks for more than half of the CU's 160 KB cannot get a neighbour.  KH_SERVE_SHARE_CU=1: the round-5 launch.
  size_t dyn_lds = DynLdsBytes(p.ll_cols);
  if (o->num_streams <= NumCUs() && !(getenv("KH_SERVE_SHARE_CU") && atoi(getenv("KH_SERVE_SHARE_CU")) != 0)) {
    hipFuncAttributes fa;
    const void *fn = b->exact ? reinterpret_cast<const void *>(&ServeKernel<true>) : reinterpret_cast<const void *>(&ServeKernel<false>);
    KH_HIP(hipFuncGetAttributes(&fa, fn));
    const size_t want_total = 88 * 1024;
    if (fa.sharedSizeBytes + dyn_lds < want_total) dyn_lds = want_total - fa.sharedSizeBytes;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dyn_lds));
  }
#ifdef KH_SERVE_MARKERS
  {
    int32_t *mark = static_cast<int32_t *>(ctl_dev);
    KH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_serve_mark), &mark, sizeof(mark)));
    if (g_wave_mark_host == nullptr) {
      KH_HIP(hipHostMalloc(reinterpret_cast<void **>(&g_wave_mark_host), sizeof(int32_t) * 64 * 1024, hipHostMallocMapped));
      memset(g_wave_mark_host, 0, sizeof(int32_t) * 64 * 1024);
    }
    int32_t *wm_dev = nullptr;
    KH_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&wm_dev), g_wave_mark_host, 0));
    if (o->num_streams > 1024) wm_dev = nullptr;
    KH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wave_mark), &wm_dev, sizeof(wm_dev)));
  }
#endif
  if (b->exact) {
    if (getenv("KH_DECODER_ORDER_SORT") != nullptr && atoi(getenv("KH_DECODER_ORDER_SORT")) != 0) p.exact_order = 2;
    hipLaunchKernelGGL(ServeKernel<true>, dim3(static_cast<unsigned>(o->num_streams)), dim3(NT), dyn_lds, o->serve_stream,
                       b->d_slots, o->d_states, static_cast<ServeCtl *>(ctl_dev), static_cast<int32_t *>(quit_dev), o->serve_ll,
                       o->serve_rows, o->serve_stride, p, idle_ticks, o->d_serve_act, (const UttX *)b->d_slotsx);
  } else {
    hipLaunchKernelGGL(ServeKernel<false>, dim3(static_cast<unsigned>(o->num_streams)), dim3(NT), dyn_lds, o->serve_stream,
                       b->d_slots, o->d_states, static_cast<ServeCtl *>(ctl_dev), static_cast<int32_t *>(quit_dev), o->serve_ll,
                       o->serve_rows, o->serve_stride, p, idle_ticks, o->d_serve_act, (const UttX *)nullptr);
  }
  KH_LAUNCH_CHECK();
  o->serve_launched = true;
  return KH_OK;
}

int kh_online_decoder_serve_start(KhOnlineDecoder *o, const float *loglikes, int ll_stride, int64_t rows_per_stream,
                                  const int32_t *tid2pdf) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(o && loglikes && ll_stride > 0 && rows_per_stream >= o->max_frames);
  if (o->num_streams > KH_WG_PER_CU * NumCUs()) {
    SetError("kh_online_decoder_serve_start: %d streams, but only %d workgroups can be resident at once", o->num_streams,
             KH_WG_PER_CU * NumCUs());
    return KH_EINVAL;
  }
  if ((rc = kh_online_decoder_serve_stop(o))) return rc;
  if (!o->serve_ctl) {
    KH_HIP(hipHostMalloc(reinterpret_cast<void **>(&o->serve_ctl), sizeof(ServeCtl) * o->num_streams, hipHostMallocMapped));
    KH_HIP(hipHostMalloc(reinterpret_cast<void **>(&o->serve_quit), 64, hipHostMallocMapped));
    KH_HIP(hipStreamCreateWithFlags(&o->serve_stream, hipStreamNonBlocking));
    o->d_serve_act = static_cast<long long *>(PoolMalloc(sizeof(long long) * o->num_streams));
    if (!o->d_serve_act) {
      SetError("kh_online_decoder_serve_start: out of device memory");
      return KH_ENOMEM;
    }
    KH_HIP(hipMemset(o->d_serve_act, 0, sizeof(long long) * o->num_streams));
    memset(o->serve_ctl, 0, sizeof(ServeCtl) * o->num_streams);
    o->serve_seq.assign(o->num_streams, 0);
  }
  for (int s = 0; s < o->num_streams; s++) {   // pick up where the launch-per-job calls left the streams
    o->serve_ctl[s].avail = o->frames[s];
    o->serve_ctl[s].decoded = o->frames[s];
    o->serve_ctl[s].ok = 1;
    // the arena's high-water mark as the LAST serving kernel reported it is stale once launch-per-job calls have decoded
    // on the stream in between (they raise SlotState::tok_hw, not this copy): 0 = the next InitDecoding is a plain kCmdInit,
    // the device clears [0, its own tok_hw) and reports the mark afresh (a host-side fill of the stale range
    // + kCmdInitCleared left finite costs behind)
    o->serve_ctl[s].hw = 0;
  }
  o->serve_ll = loglikes;
  o->serve_stride = ll_stride;
  o->serve_rows = rows_per_stream;
  o->serve_map = tid2pdf;
  return ServeEnsureRunning(o);
}

int kh_online_decoder_serve_stop(KhOnlineDecoder *o) {
  KH_CHECK_ARG(o);
  if (!o->serve_ctl || !o->serve_launched) return KH_OK;
  __atomic_store_n(o->serve_quit, 1, __ATOMIC_RELEASE);
  const int rc = ServeJoin(o, "kh_online_decoder_serve_stop");
  if (rc) return rc;
  for (int s = 0; s < o->num_streams; s++) {
    o->frames[s] = __atomic_load_n(&o->serve_ctl[s].decoded, __ATOMIC_ACQUIRE);
    o->lat_key[s] = -1;
  }
  return KH_OK;
}

static int ServeCommand(KhOnlineDecoder *o, const int32_t *streams, int n, int op_all, const int32_t *ops = nullptr) {
  for (int i = 0; i < n; i++) {
    ServeCtl &c = o->serve_ctl[streams[i]];
    const int op = ops ? ops[i] : op_all;
    if (op == kCmdInit || op == kCmdInitCleared) __atomic_store_n(&c.avail, 0, __ATOMIC_RELEASE);
    __atomic_store_n(&c.cmd_op, op, __ATOMIC_RELEASE);
    __atomic_store_n(&c.cmd_seq, ++o->serve_seq[streams[i]], __ATOMIC_RELEASE);
  }
  return ServeEnsureRunning(o);
}

// InitDecoding of the listed streams by the serving kernel (asynchronous: kh_online_decoder_serve_wait).
int kh_online_decoder_serve_init(KhOnlineDecoder *o, const int32_t *streams, int n) {
  KH_CHECK_ARG(o && o->serve_ctl && streams && n > 0 && n <= o->num_streams && DistinctStreams(o, streams, n));
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    if (o->serve_seq[s] != __atomic_load_n(&o->serve_ctl[s].ack_seq, __ATOMIC_ACQUIRE)) {
      SetError("kh_online_decoder_serve_init: stream %d still has a command in flight", s);
      return KH_ESTATE;
    }
  }
  // The reset of a large token arena by a fill kernel over the whole chip instead of by the stream's one workgroup - only
  // for a stream whose utterance is finalized and acknowledged (its workgroup waits; nobody else touches the arena).
  std::vector<int32_t> ops(n, kCmdInit);
  bool filled = false;
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    const int hw = __atomic_load_n(&o->serve_ctl[s].hw, __ATOMIC_ACQUIRE);
    // (one workgroup per CU at most: the fill kernel needs room next to the resident workgroups)
    if (o->num_streams <= NumCUs() && o->inited[s] && o->finalized[s] && hw >= (1 << 18) && hw <= o->base->h_slots[s].tok_cap) {
      hipLaunchKernelGGL(FillU32, dim3(256), dim3(256), 0, Stream(), (uint32_t *)o->base->h_slots[s].tok_cost.p, static_cast<size_t>(hw), kEncInf);
      KH_LAUNCH_CHECK();
      ops[i] = kCmdInitCleared;
      filled = true;
    }
  }
  if (filled) KH_HIP(hipStreamSynchronize(Stream()));
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    o->inited[s] = 1;
    o->finalized[s] = 0;
    o->frames[s] = 0;
    o->lat_key[s] = -1;
  }
  return ServeCommand(o, streams, n, kCmdInit, ops.data());
}

// Scores of frames [0, avail[i]) of stream streams[i] are in its score buffer (the kernels that wrote them have completed):
// the stream decodes up to there.
int kh_online_decoder_serve_publish(KhOnlineDecoder *o, const int32_t *streams, int n, const int32_t *avail) {
  KH_CHECK_ARG(o && o->serve_ctl && streams && avail && n > 0 && n <= o->num_streams);
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    KH_CHECK_ARG(s >= 0 && s < o->num_streams && avail[i] >= 0 && avail[i] <= o->max_frames);
    if (!o->inited[s] || o->finalized[s]) {
      SetError("kh_online_decoder_serve_publish: stream %d: call InitDecoding() first, and not after FinalizeDecoding()", s);
      return KH_ESTATE;
    }
    __atomic_store_n(&o->serve_ctl[s].avail, avail[i], __ATOMIC_RELEASE);
    o->lat_key[s] = -1;
  }
  return ServeEnsureRunning(o);
}

// FinalizeDecoding of the listed streams once their published frames are decoded (asynchronous).
int kh_online_decoder_serve_finalize(KhOnlineDecoder *o, const int32_t *streams, int n) {
  KH_CHECK_ARG(o && o->serve_ctl && streams && n > 0 && n <= o->num_streams && DistinctStreams(o, streams, n));
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    if (!o->inited[s] || o->finalized[s]) {
      SetError("kh_online_decoder_serve_finalize: stream %d is not in a decoding run", s);
      return KH_ESTATE;
    }
    o->finalized[s] = 1;
    o->lat_key[s] = -1;
  }
  return ServeCommand(o, streams, n, kCmdFinalize);
}

// Progress of the listed streams: NumFramesDecoded() so far, and whether a command (InitDecoding / FinalizeDecoding) is
// still in flight.  Either output may be NULL.
int kh_online_decoder_serve_poll(KhOnlineDecoder *o, const int32_t *streams, int n, int32_t *decoded, int32_t *in_flight) {
  KH_CHECK_ARG(o && o->serve_ctl && streams && n > 0);
  bool work = false, lost = false;
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    KH_CHECK_ARG(s >= 0 && s < o->num_streams);
    const ServeCtl &c = o->serve_ctl[s];
    const int ack = __atomic_load_n(&c.ack_seq, __ATOMIC_ACQUIRE);
    const int dec = __atomic_load_n(&c.decoded, __ATOMIC_ACQUIRE);
    if (!__atomic_load_n(&c.ok, __ATOMIC_ACQUIRE))
      return StreamFailed("serving kernel", s, __atomic_load_n(&c.pad1[1], __ATOMIC_ACQUIRE), dec);
    const bool pending = ack != o->serve_seq[s];
    // (between an InitDecoding command and its acknowledgement `decoded` still belongs to the previous utterance)
    o->frames[s] = (pending && (c.cmd_op == kCmdInit || c.cmd_op == kCmdInitCleared)) ? 0 : dec;
    if (decoded) decoded[i] = o->frames[s];
    if (in_flight) in_flight[i] = pending ? 1 : 0;
    const bool has_work = pending || (!o->finalized[s] && __atomic_load_n(&c.avail, __ATOMIC_ACQUIRE) > dec);
    work |= has_work;
    // a stream with work whose workgroup has left while the kernel is neither leaving nor gone: the protocol's invariant
    // (workgroups leave only together) is broken - stop the grid and start it again rather than wait for ever
    // (phase 9 is written by a workgroup of THIS launch only: ServeEnsureRunning resets the field before it launches)
    lost |= has_work && o->serve_launched && __atomic_load_n(&c.alive, __ATOMIC_ACQUIRE) == 0 && __atomic_load_n(o->serve_quit, __ATOMIC_ACQUIRE) == 0 &&
            __atomic_load_n(&c.hb_phase, __ATOMIC_ACQUIRE) == 9;
  }
  if (lost) {
    __atomic_store_n(o->serve_quit, 1, __ATOMIC_RELEASE);
    const int rc = ServeJoin(o, "kh_online_decoder_serve_poll");
    if (rc) return rc;
  }
  // a caller that only polls must still get its streams served after the grid has left by its idle time
  if (work && (!o->serve_launched || __atomic_load_n(o->serve_quit, __ATOMIC_ACQUIRE) != 0)) return ServeEnsureRunning(o);
  return KH_OK;
}

// Blocks until the listed streams have decoded everything published to them and acknowledged their commands
// (timeout_ms <= 0: 60 s).  The getters of the online decoder may be used on them afterwards.
int kh_online_decoder_serve_wait(KhOnlineDecoder *o, const int32_t *streams, int n, int timeout_ms) {
  KH_CHECK_ARG(o && o->serve_ctl && streams && n > 0);
  const auto t0 = std::chrono::steady_clock::now();
  const double limit = timeout_ms > 0 ? timeout_ms : ServeTimeoutMs();
  std::vector<int32_t> dec(n), fl(n);
  for (;;) {
    int rc = ServeEnsureRunning(o);
    if (rc) return rc;
    if ((rc = kh_online_decoder_serve_poll(o, streams, n, dec.data(), fl.data()))) return rc;
    bool all = true;
    for (int i = 0; i < n && all; i++) {
      const ServeCtl &c = o->serve_ctl[streams[i]];
      all = !fl[i] && (o->finalized[streams[i]] || dec[i] >= __atomic_load_n(&c.avail, __ATOMIC_ACQUIRE));
    }
    if (all) return KH_OK;
    if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > limit) {
      SetError("kh_online_decoder_serve_wait: timed out after %.0f ms [%s]", limit, ServeDump(o, streams, n).c_str());
      return KH_ETIMEOUT;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

// Snapshot of a stream's raw lattice into base->lats[stream] (GetRawLattice,
// lattice-faster-online-decoder.cc:143-233: before FinalizeDecoding the final costs are
// computed on the fly when use_final_probs is set, :160-165).
static int SnapshotLattice(KhOnlineDecoder *o, int stream, int use_final_probs) {
  KhDecoder *b = o->base;
  KH_CHECK_ARG(stream >= 0 && stream < o->num_streams);
  if (!o->inited[stream]) {
    SetError("stream %d: InitDecoding() has not been called", stream);
    return KH_ESTATE;
  }
  if (o->finalized[stream] && !use_final_probs) {
    SetError("You cannot call FinalizeDecoding() and then call GetRawLattice() with use_final_probs == false "
             "(lattice-faster-online-decoder.cc:156-158)");
    return KH_ESTATE;
  }
  if (o->frames[stream] <= 0) {
    SetError("stream %d: no frames decoded yet (lattice-faster-online-decoder.cc:171)", stream);
    return KH_ESTATE;
  }
  const long long key = (static_cast<long long>(o->frames[stream]) << 2) | (use_final_probs ? 2 : 0) | (o->finalized[stream] ? 1 : 0);
  if (o->lat_key[stream] == key) return KH_OK;
  hipStream_t st = Stream();
  long long pool_tok = 65536 + 512ll * o->frames[stream], pool_link = 131072 + 1024ll * o->frames[stream];
  UttOut q;
  for (int attempt = 0;; attempt++) {
    int rc = EnsurePool(b, pool_tok, pool_link);
    if (rc) return rc;
    KH_HIP(hipMemsetAsync(b->d_used, 0, sizeof(unsigned long long) * 4, st));
    std::vector<Job> jobs(1);
    jobs[0].slot = stream;
    jobs[0].op = kJobExport;
    jobs[0].ll = (GP(const float))nullptr;
    jobs[0].ll_stride = 0;
    jobs[0].n_frames = 0;
    rc = LaunchJobs(o, jobs, 0, nullptr);
    if (rc) return rc;
    unsigned long long used[4] = {0, 0, 0, 0};
    KH_HIP(hipMemcpyAsync(&q, b->d_out, sizeof(UttOut), hipMemcpyDeviceToHost, st));
    KH_HIP(hipMemcpyAsync(used, b->d_used, sizeof(used), hipMemcpyDeviceToHost, st));
    KH_HIP(hipStreamSynchronize(st));
    if (q.stats.status == 6 && attempt == 0) {  // pool too small: the exact size is known now
      pool_tok = q.n_tok + 1024;
      pool_link = q.n_link + 1024;
      continue;
    }
    if (q.stats.status != 0) {
      SetError("stream %d: decoder failed (code %d: %s)", stream, q.stats.status, StatusText(q.stats.status));
      return KH_ECAPACITY;
    }
    b->rounds.clear();
    rc = FetchPool(b, used, pool_tok, pool_link, st);
    if (rc) return rc;
    break;
  }
  b->h_round[stream] = 0;
  b->h_T[stream] = o->frames[stream];
  KhDecoder::HostPool &hp = b->rounds[0];
  if (!o->finalized[stream]) {
    // ComputeFinalCosts :309 (:860-900) on the current last frame
    const float inf = std::numeric_limits<float>::infinity();
    bool any_final = false;
    float best = inf, best_final = inf;
    (void)best; (void)best_final;
    if (use_final_probs)
      for (int k = 0; k < q.n_tok; k++)
        if (hp.t_frame[q.tok_off + k] == o->frames[stream] && b->fst->final_host[hp.t_state[q.tok_off + k]] != inf) any_final = true;
    q.stats.reached_final = any_final ? 1 : 0;
  }
  b->h_out[stream] = q;
  b->lats[stream] = KhDecoder::Lat();
  int rc = BuildLattice(b, stream);
  b->rounds.clear();
  if (rc) return rc;
  o->lat_key[stream] = key;
  return KH_OK;
}

int kh_online_decoder_get_stats(KhOnlineDecoder *o, int stream, int use_final_probs, KhDecodeStats *stats) {
  KH_CHECK_ARG(o && stats);
  int rc = SnapshotLattice(o, stream, use_final_probs);
  if (rc) return rc;
  *stats = o->base->h_out[stream].stats;
  stats->num_tokens = static_cast<int32_t>(o->base->lats[stream].state_frame.size());
  stats->num_links = static_cast<int32_t>(o->base->lats[stream].arc_src.size());
  return KH_OK;
}

int kh_online_decoder_get_raw_lattice(KhOnlineDecoder *o, int stream, int use_final_probs, int32_t *state_frame,
                                      int32_t *state_hclg, float *state_final, int32_t *arc_src, int32_t *arc_dst,
                                      int32_t *arc_ilabel, int32_t *arc_olabel, float *arc_graph,
                                      float *arc_acoustic) {
  KH_CHECK_ARG(o);
  int rc = SnapshotLattice(o, stream, use_final_probs);
  if (rc) return rc;
  return kh_decoder_get_raw_lattice(o->base, stream, state_frame, state_hclg, state_final, arc_src, arc_dst,
                                    arc_ilabel, arc_olabel, arc_graph, arc_acoustic);
}

// GetBestPath (:107-108): the reference traces token backpointers (BestPathEnd /
// TraceBackBestPath); here it is the shortest path of the same raw lattice, which
// TestGetBestPath (:113) checks to be equivalent.
int kh_online_decoder_get_best_path(KhOnlineDecoder *o, int stream, int use_final_probs, int32_t *alignment,
                                    int cap_ali, int32_t *n_ali, int32_t *words, int cap_words, int32_t *n_words,
                                    float *graph_cost, float *acoustic_cost) {
  KH_CHECK_ARG(o);
  int rc = SnapshotLattice(o, stream, use_final_probs);
  if (rc) return rc;
  return kh_decoder_get_best_path(o->base, stream, alignment, cap_ali, n_ali, words, cap_words, n_words, graph_cost,
                                  acoustic_cost);
}

}  // extern "C"
