// kh_determinize.hip — pruned lattice determinization on the host.
//
// Replaces DeterminizeLatticePhonePrunedWrapper (lat/determinize-lattice-pruned.cc:1497-1519, called from
// DecodeUtteranceLatticeFaster, decoder/decoder-wrappers.cc:264-274) with its whole option set
// (DeterminizeLatticePhonePrunedOptions, lat/determinize-lattice-pruned.h:145-175):
//   * phone_determinize (default true): a first pass on phone + word labels (:1386-1404) — phone labels are put on the
//     arcs whose transition-id leaves HMM-state 0 and is no self-loop (:1310-1360), the lattice is determinized on the
//     joint labels and written back as a state-level lattice, the phone labels are removed again;
//   * word_determinize (default true): the pass on word labels (:1202-1260) -> CompactLattice;
//   * minimize (default false): PushCompactLatticeStrings + PushCompactLatticeWeights (lat/push-lattice.cc) and
//     MinimizeCompactLattice (lat/minimize-lattice.cc);
//   * delta, max_mem; the retry with a narrower beam on a pruned lattice when the memory limit stopped a pass before
//     half of the beam was reached (:1211-1241, kaldi::PruneLattice lat/lattice-functions.cc:187-266).
// The raw state-level lattice of the decoder -> a CompactLattice that is deterministic on WORDS, every word sequence
// keeping its best path (LatticeWeight order, fstext/lattice-weight.h:297-312, then the string order of
// determinize-lattice-pruned.cc:619-643) with that path's transition-id string, pruned to `beam`.
//
// Host code, as in the reference (utterances are independent: kh_decoder_decode runs it on the threads that drain the
// kernel's completion list).  The algorithm is the reference's — weighted subset construction in the semiring
// (LatticeWeight x transition-id string), tasks = (output state, label) expansions taken from a priority queue in order
// of their best complete-path cost and dropped beyond best + beam, subsets normalised by their best weight and longest
// common string prefix, a look-aside table from INITIAL subsets (before the epsilon closure) to output states — and so
// are its results, to the state: the restatement under oracle/determinize_oracle.cc is the differential oracle
// (tests/test_determinize_oracle.py: same number of states and arcs, the same weighted language and alignments).  That
// includes one property of this version of the reference: an output state is shared only through the initial-subset
// table (MinimalToStateId :528-557 finds an equal minimal subset but creates a new state all the same), which prunes per
// PATH — a state reached again on a worse path expands only what is inside the beam on that path.
// KH_DETERMINIZE_SHARE_MINIMAL=1 shares states by minimal subset as later Kaldi versions do.
//
// What is NOT the reference's is the machinery, built for this being the CPU tail behind a decoder that is three orders
// of magnitude faster than the CPU one: transition-id strings live in a flat trie with an open-addressing table (one
// probe per appended label; the reference's repository is a node-per-entry hash set), a common prefix is the lowest
// common ancestor of two trie nodes and removing it costs the length of what REMAINS (the reference converts every string
// to a vector: its whole length), the epsilon closure runs over the lattice in topological order with a heap of state
// ranks (every state is settled once; the reference's queue revisits), and all scratch memory belongs to a per-thread
// workspace that is reused from utterance to utterance (no allocation per lattice: with dozens of host threads the
// process's memory-map lock was the bottleneck).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <queue>
#include <chrono>
#include <unordered_map>
#include <vector>

#include "kh_common.h"

using namespace kh;

struct KhCompactLattice {
  std::vector<int32_t> arc_src, arc_dst, arc_label, arc_str_off, strings;
  std::vector<float> arc_g, arc_a, final_g, final_a;
  std::vector<int32_t> final_str_off;  // [n_states + 1] into final_strings
  std::vector<int32_t> final_strings;
  int32_t n_states = 0;
  int complete = 1;
};

namespace {

const float kInfF = std::numeric_limits<float>::infinity();
const double kInfD = std::numeric_limits<double>::infinity();

struct LW { float g, a; };  // LatticeWeight (value1 = graph, value2 = acoustic)
inline LW Times(LW x, LW y) { return LW{x.g + y.g, x.a + y.a}; }
inline bool IsZero(LW x) { return x.g == kInfF && x.a == kInfF; }
inline LW Divide(LW x, LW y) {  // lattice-weight.h:343-361
  const float a = x.g - y.g, b = x.a - y.a;
  if (a != a || b != b || a == -kInfF || b == -kInfF || a == kInfF || b == kInfF) return LW{kInfF, kInfF};
  return LW{a, b};
}
inline double Cost(LW x) { return static_cast<double>(x.g) + static_cast<double>(x.a); }  // ConvertToCost lattice-weight.h:794-806
// fstext/lattice-weight.h:297-312: 1 if x is better (smaller) than y
inline int CompareW(LW x, LW y) {
  const float f1 = x.g + x.a, f2 = y.g + y.a;
  if (f1 < f2) return 1;
  if (f1 > f2) return -1;
  if (x.g < y.g) return 1;
  if (x.g > y.g) return -1;
  return 0;
}
inline bool ApproxEqualW(LW x, LW y, float delta) {  // lattice-weight.h:364-370
  if (x.g == y.g && x.a == y.a) return true;
  return std::fabs((x.g + x.a) - (y.g + y.a)) <= delta;
}

// ---------------------------------------------------------------- transition-id strings
// A string is a node of a trie (node 0 = the empty string); equal strings are the same node.
struct Strings {
  struct Node { int32_t parent, label, depth; };
  std::vector<Node> nodes;
  std::vector<uint64_t> keys;   // open addressing: (parent << 32 | label) + 1, 0 = empty
  std::vector<int32_t> vals;
  uint32_t mask = 0;
  std::vector<int32_t> tmp;
  void Reset(size_t expected) {
    nodes.clear();
    nodes.push_back(Node{-1, 0, 0});
    size_t cap = 1024;
    while (cap < 4 * expected) cap <<= 1;
    if (keys.size() != cap) { keys.assign(cap, 0); vals.assign(cap, 0); }
    else std::fill(keys.begin(), keys.end(), 0);
    mask = static_cast<uint32_t>(cap - 1);
  }
  static uint32_t Hash(uint64_t k) {
    k *= 0x9E3779B97F4A7C15ull;
    return static_cast<uint32_t>(k >> 32);
  }
  void Grow() {
    std::vector<uint64_t> ok;
    std::vector<int32_t> ov;
    ok.swap(keys);
    ov.swap(vals);
    keys.assign(ok.size() * 2, 0);
    vals.assign(ok.size() * 2, 0);
    mask = static_cast<uint32_t>(keys.size() - 1);
    for (size_t i = 0; i < ok.size(); i++) {
      if (!ok[i]) continue;
      uint32_t h = Hash(ok[i]) & mask;
      while (keys[h]) h = (h + 1) & mask;
      keys[h] = ok[i];
      vals[h] = ov[i];
    }
  }
  int32_t Successor(int32_t s, int32_t label) {
    const uint64_t key = ((static_cast<uint64_t>(static_cast<uint32_t>(s)) << 32) | static_cast<uint32_t>(label)) + 1;
    uint32_t h = Hash(key) & mask;
    while (keys[h]) {
      if (keys[h] == key) return vals[h];
      h = (h + 1) & mask;
    }
    const int32_t id = static_cast<int32_t>(nodes.size());
    nodes.push_back(Node{s, label, nodes[s].depth + 1});
    keys[h] = key;
    vals[h] = id;
    if (nodes.size() * 2 > keys.size()) Grow();
    return id;
  }
  int Depth(int32_t s) const { return nodes[s].depth; }
  int32_t CommonPrefix(int32_t a, int32_t b) const {
    while (nodes[a].depth > nodes[b].depth) a = nodes[a].parent;
    while (nodes[b].depth > nodes[a].depth) b = nodes[b].parent;
    while (a != b) { a = nodes[a].parent; b = nodes[b].parent; }
    return a;
  }
  void ToVector(int32_t s, std::vector<int32_t> *v) const {
    v->resize(nodes[s].depth);
    for (int i = nodes[s].depth - 1; i >= 0; i--) { (*v)[i] = nodes[s].label; s = nodes[s].parent; }
  }
  // the string s without its first n labels: costs what remains, not the whole string
  int32_t RemovePrefix(int32_t s, int n) {
    if (n == 0) return s;
    const int r = nodes[s].depth - n;
    tmp.resize(r);
    for (int i = r - 1; i >= 0; i--) { tmp[i] = nodes[s].label; s = nodes[s].parent; }
    int32_t out = 0;
    for (int i = 0; i < r; i++) out = Successor(out, tmp[i]);
    return out;
  }
  // The same for the elements of ONE subset, all losing the same n labels (Normalize): the strings of a subset share
  // trie paths below the common prefix (its elements were reached along common arcs), so the image of every node visited
  // is remembered for the duration of the call (BeginRemove) and a string costs only the nodes no earlier element has
  // brought over: 19 insertions per element on the benchmark's raw lattices without it.  New nodes are created in the same
  // order as by RemovePrefix element after element, so the node ids - and everything that follows - are unchanged.
  std::vector<int32_t> memo_new;
  std::vector<uint32_t> memo_epoch;
  uint32_t epoch = 0;
  void BeginRemove() {
    if (++epoch == 0) { std::fill(memo_epoch.begin(), memo_epoch.end(), 0u); epoch = 1; }
  }
  int32_t RemovePrefixShared(int32_t s, int n) {
    if (n == 0) return s;
    tmp.clear();
    int32_t cur = s, base = 0;
    while (nodes[cur].depth > n) {
      if (static_cast<size_t>(cur) < memo_epoch.size() && memo_epoch[cur] == epoch) { base = memo_new[cur]; break; }
      tmp.push_back(cur);
      cur = nodes[cur].parent;
    }
    for (size_t i = tmp.size(); i-- > 0;) {
      const int32_t old = tmp[i];
      base = Successor(base, nodes[old].label);
      if (static_cast<size_t>(old) >= memo_epoch.size()) {
        memo_epoch.resize(nodes.size() + nodes.size() / 2 + 64, 0u);
        memo_new.resize(memo_epoch.size());
      }
      memo_new[old] = base;
      memo_epoch[old] = epoch;
    }
    return base;
  }
  int32_t Concatenate(int32_t a, int32_t b) {
    if (a == 0) return b;
    if (b == 0) return a;
    std::vector<int32_t> v;
    ToVector(b, &v);
    for (int32_t x : v) a = Successor(a, x);
    return a;
  }
  // determinize-lattice-pruned.cc:619-643 on two strings: the longer one is "less" (-1), the shorter "more" (1); equal
  // lengths: at the first position where they differ the smaller label is "less"
  int Compare(int32_t a, int32_t b) const {
    if (a == b) return 0;
    if (nodes[a].depth > nodes[b].depth) return -1;
    if (nodes[a].depth < nodes[b].depth) return 1;
    int32_t la = 0, lb = 0;
    while (a != b) {   // (climbing from the ends: the last differing pair seen is the first position where they differ)
      la = nodes[a].label;
      lb = nodes[b].label;
      a = nodes[a].parent;
      b = nodes[b].parent;
    }
    return la < lb ? -1 : (la > lb ? 1 : 0);
  }
  size_t MemSize() const { return (nodes.size() - 1) * 16 * 2; }   // LatticeStringRepository::MemSize: entries x sizeof(Entry) x 2
  // LatticeStringRepository::Rebuild (determinize-lattice-inl.h:188-201): keep the marked strings and their prefixes, drop
  // the rest; new_id[old] = the node's new number (-1: dropped).  A node's parent was created before it, so one pass in
  // order renumbers parents before children.
  void Rebuild(const std::vector<char> &keep, std::vector<int32_t> *new_id) {
    new_id->assign(nodes.size(), -1);
    (*new_id)[0] = 0;
    size_t m = 1;
    for (size_t i = 1; i < nodes.size(); i++) {
      if (!keep[i]) continue;
      (*new_id)[i] = static_cast<int32_t>(m);
      nodes[m] = Node{(*new_id)[nodes[i].parent], nodes[i].label, nodes[i].depth};
      m++;
    }
    nodes.resize(m);
    std::fill(keys.begin(), keys.end(), 0);
    for (size_t i = 1; i < m; i++) {
      const uint64_t key = ((static_cast<uint64_t>(static_cast<uint32_t>(nodes[i].parent)) << 32) | static_cast<uint32_t>(nodes[i].label)) + 1;
      uint32_t h = Hash(key) & mask;
      while (keys[h]) h = (h + 1) & mask;
      keys[h] = key;
      vals[h] = static_cast<int32_t>(i);
    }
  }
};

// ---------------------------------------------------------------- lattices
// State-level lattice the passes read: CSR by state, states in topological order, a state's arcs sorted by label
// (label 0 first: the closure stops at the first labelled arc).  label = what the pass determinizes on (word, or phone
// / word); tid = the transition-id (0 = none).
struct Lat {
  int32_t n = 0;
  std::vector<int64_t> off;
  std::vector<int32_t> label, tid, next;
  std::vector<LW> w;
  std::vector<LW> fin;          // Zero = not final
  void Clear() { n = 0; off.clear(); label.clear(); tid.clear(); next.clear(); w.clear(); fin.clear(); }
};

struct Elem { int32_t state; int32_t str; LW w; };
struct TempArc { int32_t label, str, next; LW w; };   // next == -1: a final weight

struct Options {
  float delta = 0.0009765625f;
  int64_t max_mem = -1;
  float retry_cutoff = 0.5f;
  bool share_minimal = false;
};

// One determinization pass (LatticeDeterminizerPruned).  The object is a per-thread workspace: every container keeps its
// capacity from one lattice to the next.
struct Pass {
  const Lat *in = nullptr;
  double beam = 0.0, cutoff = 0.0;
  Options opts;
  Strings strs;
  std::vector<double> backward;
  std::vector<char> has_label_or_final;
  // output states
  struct OutState { int32_t sub_b, sub_e; double forward_cost; };
  std::vector<OutState> out;
  std::vector<Elem> subsets;           // the minimal subsets of the output states, concatenated
  std::vector<std::vector<TempArc>> out_arcs;
  long long num_arcs = 0, num_elems = 0;
  // tasks
  struct Task { int32_t state, label, sub_b, sub_e; double priority; };
  std::vector<Task> tasks;             // every task created (its subset in task_elems)
  std::vector<Elem> task_elems;
  struct TaskWorse { const std::vector<Task> *t; bool operator()(int32_t x, int32_t y) const { return (*t)[x].priority > (*t)[y].priority; } };
  std::vector<int32_t> heap;
  // initial-subset table: hash of the (state, string) list -> entries
  struct InitEntry { int32_t sub_b, sub_e, out_state, str; LW w; };
  std::vector<InitEntry> init_entries;
  std::vector<Elem> init_elems;
  std::unordered_multimap<uint64_t, int32_t> init_index, minimal_index;
  // scratch
  std::vector<int32_t> slot_of;        // position of a lattice state in the subset being closed, -1
  std::vector<int32_t> rank_heap;
  std::vector<Elem> closure, sub_tmp, minimal_tmp;
  struct LE { int32_t label; Elem e; };
  std::vector<LE> all;

  int CompareElem(LW w1, int32_t s1, LW w2, int32_t s2) const {
    const int c = CompareW(w1, w2);
    return c != 0 ? c : strs.Compare(s1, s2);
  }
  static uint64_t HashSubset(const Elem *b, const Elem *e) {
    uint64_t h = 1469598103934665603ull;
    for (; b != e; ++b) {
      h = (h ^ static_cast<uint32_t>(b->state)) * 1099511628211ull;
      h = (h ^ static_cast<uint32_t>(b->str)) * 1099511628211ull;
    }
    return h;
  }
  bool SameSubset(const Elem *x, const Elem *xe, const Elem *y, const Elem *ye) const {  // SubsetEqual :457-473
    if (xe - x != ye - y) return false;
    for (; x != xe; ++x, ++y)
      if (x->state != y->state || x->str != y->str || !ApproxEqualW(x->w, y->w, opts.delta)) return false;
    return true;
  }

  // EpsilonClosure :650-748.  The lattice is topologically sorted, so the states are settled in increasing order: a heap
  // of state ids, every state expanded once, with its final best (weight, string).
  void EpsilonClosure(const Elem *b, const Elem *e) {
    closure.assign(b, e);
    rank_heap.clear();
    for (size_t i = 0; i < closure.size(); i++) {
      slot_of[closure[i].state] = static_cast<int32_t>(i);
      rank_heap.push_back(closure[i].state);
    }
    std::make_heap(rank_heap.begin(), rank_heap.end(), std::greater<int32_t>());
    const Lat &L = *in;
    while (!rank_heap.empty()) {
      std::pop_heap(rank_heap.begin(), rank_heap.end(), std::greater<int32_t>());
      const int32_t s = rank_heap.back();
      rank_heap.pop_back();
      const Elem el = closure[slot_of[s]];
      for (int64_t k = L.off[s]; k < L.off[s + 1] && L.label[k] == 0; k++) {
        if (IsZero(L.w[k])) continue;
        const int32_t ns = L.next[k];
        const LW nw = Times(el.w, L.w[k]);
        const int32_t at = slot_of[ns];
        if (at < 0) {
          slot_of[ns] = static_cast<int32_t>(closure.size());
          closure.push_back(Elem{ns, L.tid[k] != 0 ? strs.Successor(el.str, L.tid[k]) : el.str, nw});
          rank_heap.push_back(ns);
          std::push_heap(rank_heap.begin(), rank_heap.end(), std::greater<int32_t>());
        } else {
          int c = CompareW(nw, closure[at].w);
          int32_t nstr = -1;
          if (c == 0) {   // "a tie on weights ... a rare case"
            nstr = L.tid[k] != 0 ? strs.Successor(el.str, L.tid[k]) : el.str;
            c = strs.Compare(nstr, closure[at].str);
          }
          if (c == 1) {   // (ns > s: not expanded yet, and in the heap already)
            if (nstr < 0) nstr = L.tid[k] != 0 ? strs.Successor(el.str, L.tid[k]) : el.str;
            closure[at].w = nw;
            closure[at].str = nstr;
          }
        }
      }
    }
    for (const Elem &x : closure) slot_of[x.state] = -1;
  }
  // ConvertToMinimal :508-525 on `closure`, left sorted by state (the order the reference's subset map gives)
  void ClosureToMinimal() {
    size_t m = 0;
    for (size_t i = 0; i < closure.size(); i++)
      if (has_label_or_final[closure[i].state]) closure[m++] = closure[i];
    closure.resize(m);
    std::sort(closure.begin(), closure.end(), [](const Elem &x, const Elem &y) { return x.state < y.state; });
  }

  // NormalizeSubset :796-824 on [b, e) in place
  void Normalize(Elem *b, Elem *e, LW *tot, int32_t *common) {
    if (b == e) { *tot = LW{kInfF, kInfF}; *common = 0; return; }
    LW best = b->w;
    int32_t pre = b->str;
    for (Elem *x = b + 1; x != e; ++x) {
      if (CompareW(x->w, best) == 1) best = x->w;   // Plus(weight, x): the better of the two, the first on a tie
      if (pre != 0) pre = strs.CommonPrefix(pre, x->str);   // (the empty string stays the common prefix: no walk)
    }
    const int n_pre = strs.Depth(pre);
    strs.BeginRemove();
    for (Elem *x = b; x != e; ++x) {
      x->w = Divide(x->w, best);
      x->str = strs.RemovePrefixShared(x->str, n_pre);
    }
    *tot = best;
    *common = pre;
  }

  int32_t NewOutState(const Elem *b, const Elem *e, double forward_cost) {
    const int32_t id = static_cast<int32_t>(out.size());
    const int32_t sb = static_cast<int32_t>(subsets.size());
    subsets.insert(subsets.end(), b, e);
    out.push_back(OutState{sb, static_cast<int32_t>(subsets.size()), forward_cost});
    if (out_arcs.size() <= static_cast<size_t>(id)) out_arcs.resize(2 * id + 64);
    out_arcs[id].clear();
    num_elems += e - b;
    ProcessFinal(id);
    ProcessTransitions(id);
    return id;
  }

  // MinimalToStateId :528-557 ([b, e) must not point into `subsets`)
  int32_t MinimalToStateId(const Elem *b, const Elem *e, double forward_cost) {
    if (opts.share_minimal) {
      const uint64_t h = HashSubset(b, e);
      auto range = minimal_index.equal_range(h);
      for (auto it = range.first; it != range.second; ++it) {
        const OutState &S = out[it->second];
        if (SameSubset(subsets.data() + S.sub_b, subsets.data() + S.sub_e, b, e)) return it->second;
      }
      minimal_index.emplace(h, static_cast<int32_t>(out.size()));
    }
    return NewOutState(b, e, forward_cost);
  }

  // InitialToStateId :561-609: [b, e) = a normalised initial subset (sorted by state, one element per state)
  int32_t InitialToStateId(const Elem *b, const Elem *e, double forward_cost, LW *remaining, int32_t *common) {
    const uint64_t h = HashSubset(b, e);
    auto range = init_index.equal_range(h);
    for (auto it = range.first; it != range.second; ++it) {
      const InitEntry &I = init_entries[it->second];
      if (SameSubset(init_elems.data() + I.sub_b, init_elems.data() + I.sub_e, b, e)) {
        *remaining = I.w;
        *common = I.str;
        return I.out_state;
      }
    }
    InitEntry I;
    I.sub_b = static_cast<int32_t>(init_elems.size());
    init_elems.insert(init_elems.end(), b, e);
    I.sub_e = static_cast<int32_t>(init_elems.size());
    EpsilonClosure(b, e);
    ClosureToMinimal();
    Normalize(closure.data(), closure.data() + closure.size(), &I.w, &I.str);
    forward_cost += Cost(I.w);
    minimal_tmp = closure;
    I.out_state = MinimalToStateId(minimal_tmp.data(), minimal_tmp.data() + minimal_tmp.size(), forward_cost);
    *remaining = I.w;
    *common = I.str;
    init_index.emplace(h, static_cast<int32_t>(init_entries.size()));
    init_entries.push_back(I);
    num_elems += I.sub_e - I.sub_b;
    return I.out_state;
  }

  // ProcessFinal :755-791
  void ProcessFinal(int32_t id) {
    const OutState S = out[id];
    bool have = false;
    LW bw{kInfF, kInfF};
    int32_t bs = 0;
    for (int32_t i = S.sub_b; i < S.sub_e; i++) {
      const Elem &el = subsets[i];
      const LW f = in->fin[el.state];
      if (IsZero(f)) continue;
      const LW w = Times(el.w, f);
      if (IsZero(w)) continue;
      if (!have || CompareElem(w, el.str, bw, bs) == 1) { bw = w; bs = el.str; have = true; }
    }
    if (have && Cost(bw) + S.forward_cost <= cutoff) {   // only inside the pruning beam (:776)
      out_arcs[id].push_back(TempArc{0, bs, -1, bw});
      num_arcs++;
    }
  }

  // ProcessTransitions :924-1001
  void ProcessTransitions(int32_t id) {
    const OutState S = out[id];
    const Lat &L = *in;
    all.clear();
    for (int32_t i = S.sub_b; i < S.sub_e; i++) {
      const Elem el = subsets[i];
      for (int64_t k = L.off[el.state]; k < L.off[el.state + 1]; k++) {
        if (L.label[k] == 0 || IsZero(L.w[k])) continue;
        all.push_back(LE{L.label[k], Elem{L.next[k], L.tid[k] != 0 ? strs.Successor(el.str, L.tid[k]) : el.str, Times(el.w, L.w[k])}});
      }
    }
    std::sort(all.begin(), all.end(), [](const LE &x, const LE &y) { return x.label != y.label ? x.label < y.label : x.e.state < y.e.state; });
    size_t i = 0;
    while (i < all.size()) {
      Task t;
      t.state = id;
      t.label = all[i].label;
      t.priority = kInfD;
      t.sub_b = static_cast<int32_t>(task_elems.size());
      while (i < all.size() && all[i].label == t.label) {
        const Elem &el = all[i].e;
        t.priority = std::min(t.priority, Cost(el.w) + backward[el.state]);
        // MakeSubsetUnique :829-861: one element per state, the best
        if (static_cast<int32_t>(task_elems.size()) > t.sub_b && task_elems.back().state == el.state) {
          if (CompareElem(el.w, el.str, task_elems.back().w, task_elems.back().str) == 1) { task_elems.back().w = el.w; task_elems.back().str = el.str; }
        } else {
          task_elems.push_back(el);
        }
        i++;
      }
      t.sub_e = static_cast<int32_t>(task_elems.size());
      t.priority += S.forward_cost;
      if (t.priority > cutoff) {
        task_elems.resize(t.sub_b);
      } else {
        tasks.push_back(t);
        heap.push_back(static_cast<int32_t>(tasks.size() - 1));
        std::push_heap(heap.begin(), heap.end(), TaskWorse{&tasks});
      }
    }
  }

  // ProcessTransition :870-898
  void ProcessTransition(const Task &t) {
    sub_tmp.assign(task_elems.begin() + t.sub_b, task_elems.begin() + t.sub_e);
    LW tot, next_tot;
    int32_t common, next_common;
    Normalize(sub_tmp.data(), sub_tmp.data() + sub_tmp.size(), &tot, &common);
    const double forward_cost = out[t.state].forward_cost + Cost(tot);
    const int32_t next = InitialToStateId(sub_tmp.data(), sub_tmp.data() + sub_tmp.size(), forward_cost, &next_tot, &next_common);
    out_arcs[t.state].push_back(TempArc{t.label, strs.Concatenate(common, next_common), next, Times(tot, next_tot)});
    num_arcs++;
  }

  // RebuildRepository :226-269: the strings still referred to - by the output states' subsets and arcs, by the tasks in
  // the queue, by the initial-subset table - survive; everything else the trie holds is dropped.
  void RebuildRepository() {
    std::vector<char> keep(strs.nodes.size(), 0);
    auto mark = [&](int32_t s) { while (s > 0 && !keep[s]) { keep[s] = 1; s = strs.nodes[s].parent; } };
    for (const Elem &e : subsets) mark(e.str);
    for (size_t i = 0; i < out.size(); i++) for (const TempArc &a : out_arcs[i]) mark(a.str);
    std::vector<char> live(task_elems.size(), 0);
    for (int32_t t : heap)
      for (int32_t i = tasks[t].sub_b; i < tasks[t].sub_e; i++) { live[i] = 1; mark(task_elems[i].str); }
    for (const Elem &e : init_elems) mark(e.str);
    for (const InitEntry &I : init_entries) mark(I.str);
    std::vector<int32_t> id;
    strs.Rebuild(keep, &id);
    for (Elem &e : subsets) e.str = id[e.str];
    for (size_t i = 0; i < out.size(); i++) for (TempArc &a : out_arcs[i]) a.str = id[a.str];
    for (size_t i = 0; i < task_elems.size(); i++) task_elems[i].str = live[i] ? id[task_elems[i].str] : 0;   // (processed tasks: dead)
    for (Elem &e : init_elems) e.str = id[e.str];
    for (InitEntry &I : init_entries) I.str = id[I.str];
    // the look-aside tables hash (state, string id) lists: re-key them
    init_index.clear();
    for (size_t k = 0; k < init_entries.size(); k++)
      init_index.emplace(HashSubset(init_elems.data() + init_entries[k].sub_b, init_elems.data() + init_entries[k].sub_e), static_cast<int32_t>(k));
    if (opts.share_minimal) {
      minimal_index.clear();
      for (size_t k = 0; k < out.size(); k++)
        minimal_index.emplace(HashSubset(subsets.data() + out[k].sub_b, subsets.data() + out[k].sub_e), static_cast<int32_t>(k));
    }
  }

  // CheckMemoryUsage :271-328 (sizeof(Entry) = 16, sizeof(TempArc) = 32, sizeof(Element) = 24 on the reference's 64-bit build)
  bool CheckMemoryUsage() {
    if (opts.max_mem <= 0) return true;
    const long long arcs_size = num_arcs * 32, elems_size = num_elems * 24;
    if (static_cast<long long>(strs.MemSize()) + arcs_size + elems_size > opts.max_mem) {
      RebuildRepository();
      if (static_cast<long long>(strs.MemSize()) + arcs_size + elems_size > static_cast<long long>(opts.max_mem * 0.8)) return false;
    }
    return true;
  }

  // Determinize :330-378.  Returns true when the queue was emptied; *effective_beam as the reference.
  bool Run(const Lat &lat, double beam_in, const Options &o, double *effective_beam) {
    in = &lat;
    beam = beam_in;
    opts = o;
    const int32_t n = lat.n;
    out.clear(); subsets.clear(); tasks.clear(); task_elems.clear(); heap.clear();
    init_entries.clear(); init_elems.clear(); init_index.clear(); minimal_index.clear();
    num_arcs = 0; num_elems = 0;
    strs.Reset(n > 0 ? static_cast<size_t>(lat.off[n]) + 1024 : 1024);
    slot_of.assign(n, -1);
    backward.assign(n, kInfD);
    has_label_or_final.assign(n, 0);
    // ComputeBackwardWeight :1030-1054
    for (int32_t s = n - 1; s >= 0; s--) {
      double b = Cost(lat.fin[s]);
      bool keep = !IsZero(lat.fin[s]);
      for (int64_t k = lat.off[s]; k < lat.off[s + 1]; k++) {
        b = std::min(b, Cost(lat.w[k]) + backward[lat.next[k]]);
        if (lat.label[k] != 0 && !IsZero(lat.w[k])) keep = true;
      }
      backward[s] = b;
      has_label_or_final[s] = keep ? 1 : 0;
    }
    if (n == 0) { if (effective_beam) *effective_beam = beam; return true; }
    cutoff = backward[0] + beam;
    // InitializeDeterminization :1056-1109: the start state's subset is not normalised
    {
      const Elem start{0, 0, LW{0.f, 0.f}};
      EpsilonClosure(&start, &start + 1);
      ClosureToMinimal();
      minimal_tmp = closure;
      if (opts.share_minimal) minimal_index.emplace(HashSubset(minimal_tmp.data(), minimal_tmp.data() + minimal_tmp.size()), 0);
      NewOutState(minimal_tmp.data(), minimal_tmp.data() + minimal_tmp.size(), 0.0);
    }
    bool complete = true;
    while (!heap.empty()) {
      if (out.size() % 10 == 0 && !CheckMemoryUsage()) { complete = false; break; }
      std::pop_heap(heap.begin(), heap.end(), TaskWorse{&tasks});
      const Task t = tasks[heap.back()];
      heap.pop_back();
      ProcessTransition(t);
    }
    if (effective_beam) *effective_beam = heap.empty() ? beam : tasks[heap.front()].priority - backward[0];
    return complete;
  }
};

// ---------------------------------------------------------------- building / converting lattices
// arcs with arbitrary state numbering, start = state `start`.  Produces the CSR form: states in the topological order the
// reference's wrapper would work on (below; the start state first), arcs sorted by label within a state.  Returns false on a cycle.
struct RawArc { int32_t src, label, tid, next; LW w; };
bool BuildLat(int32_t n, int32_t start, const std::vector<RawArc> &arcs, const std::vector<LW> &fin, Lat *L, bool always_renumber = false) {
  std::vector<int32_t> order, id(n, -1);
  std::vector<int64_t> off(n + 1, 0);
  for (const RawArc &a : arcs) off[a.src + 1]++;
  for (int32_t s = 0; s < n; s++) off[s + 1] += off[s];
  std::vector<int32_t> by_src(arcs.size());
  {
    std::vector<int64_t> fill(off.begin(), off.end() - 1);
    for (size_t j = 0; j < arcs.size(); j++) by_src[fill[arcs[j].src]++] = static_cast<int32_t>(j);
  }
  // The numbering DeterminizeLatticePhonePrunedWrapper works on (:1503-1512).  It asks ifst->Properties(kTopSorted) first:
  // a lattice whose every arc leads to a higher state number (a raw lattice without backward epsilon arcs) keeps its
  // numbering.  Otherwise fst/topsort.h: the REVERSE FINISHING order of a depth-first search from the start state, then
  // from every state not yet visited in state order, a state's arcs taken in the order they were given.  Any topological
  // order gives the same determinized lattice; this one also gives the reference's string repository - the epsilon
  // closure settles states in increasing number, and how many strings it interns for candidates that are improved on
  // later depends on that order - hence the same point at which max_mem stops a pass.  (Round 6: rounds 3-5 used Kahn's
  // order; tests/test_gpu_determinize.py found it stopping 50 output states away from the oracle at the memory limit.)
  // always_renumber: the two TopSort calls of DeterminizeLatticePhonePrunedFirstPass (:1410, :1417) do not ask first.
  order.assign(n, -1);
  bool sorted_already = start == 0 && !always_renumber;
  for (size_t j = 0; j < arcs.size() && sorted_already; j++) sorted_already = arcs[j].next > arcs[j].src;
  if (sorted_already) {
    for (int32_t s = 0; s < n; s++) order[s] = s;
  } else {
    std::vector<char> color(n, 0);   // 0 white, 1 grey, 2 black
    std::vector<std::pair<int32_t, int64_t>> stack;
    int32_t n_fin = 0;
    bool acyclic = true;
    auto visit = [&](int32_t root) {
      if (color[root]) return;
      color[root] = 1;
      stack.emplace_back(root, off[root]);
      while (!stack.empty()) {
        const int32_t s = stack.back().first;
        const int64_t k = stack.back().second;
        if (k < off[s + 1]) {
          stack.back().second++;
          const int32_t t = arcs[by_src[k]].next;
          if (color[t] == 0) { color[t] = 1; stack.emplace_back(t, off[t]); }
          else if (color[t] == 1) acyclic = false;
        } else {
          color[s] = 2;
          order[n - 1 - n_fin++] = s;
          stack.pop_back();
        }
      }
    };
    if (start >= 0 && start < n) visit(start);
    for (int32_t s = 0; s < n; s++) visit(s);
    if (!acyclic) return false;
  }
  for (int32_t i = 0; i < n; i++) id[order[i]] = i;
  L->Clear();
  L->n = n;
  L->off.assign(n + 1, 0);
  L->fin.resize(n);
  const size_t m = arcs.size();
  L->label.resize(m); L->tid.resize(m); L->next.resize(m); L->w.resize(m);
  int64_t pos = 0;
  std::vector<int32_t> tmp;
  for (int32_t i = 0; i < n; i++) {
    const int32_t s = order[i];
    L->fin[i] = fin[s];
    tmp.assign(by_src.begin() + off[s], by_src.begin() + off[s + 1]);
    // ArcSort on the label, stable.  A lattice state has a handful of arcs: insertion sort in place (std::stable_sort
    // takes a temporary buffer from the heap per call - one malloc per lattice state was a fifth of the whole run)
    if (tmp.size() <= 256) {
      for (size_t a = 1; a < tmp.size(); a++) {
        const int32_t v = tmp[a];
        const int32_t lv = arcs[v].label;
        size_t q = a;
        for (; q > 0 && arcs[tmp[q - 1]].label > lv; q--) tmp[q] = tmp[q - 1];
        tmp[q] = v;
      }
    } else {
      std::stable_sort(tmp.begin(), tmp.end(), [&](int32_t x, int32_t y) { return arcs[x].label < arcs[y].label; });
    }
    for (int32_t j : tmp) {
      L->label[pos] = arcs[j].label; L->tid[pos] = arcs[j].tid; L->next[pos] = id[arcs[j].next]; L->w[pos] = arcs[j].w;
      pos++;
    }
    L->off[i + 1] = pos;
  }
  return true;
}

// kaldi::PruneLattice lat/lattice-functions.cc:187-266 on a CSR lattice -> raw arcs of what survives (states renumbered)
void PruneToRaw(const Lat &L, double beam, int32_t *n_out, std::vector<RawArc> *arcs, std::vector<LW> *fin) {
  const int32_t n = L.n;
  std::vector<double> fwd(n, kInfD), bwd(n, kInfD);
  fwd[0] = 0.0;
  double best = kInfD;
  for (int32_t s = 0; s < n; s++) {
    for (int64_t k = L.off[s]; k < L.off[s + 1]; k++) fwd[L.next[k]] = std::min(fwd[L.next[k]], fwd[s] + Cost(L.w[k]));
    best = std::min(best, fwd[s] + Cost(L.fin[s]));
  }
  const double cut = best + beam;
  std::vector<char> keep_arc(L.off[n], 0), keep_fin(n, 0);
  for (int32_t s = n - 1; s >= 0; s--) {
    double b = Cost(L.fin[s]);
    if (!(b + fwd[s] > cut) && b != kInfD) keep_fin[s] = 1;
    for (int64_t k = L.off[s]; k < L.off[s + 1]; k++) {
      const double ab = Cost(L.w[k]) + bwd[L.next[k]];
      if (ab < b) b = ab;
      if (!(fwd[s] + ab > cut)) keep_arc[k] = 1;
    }
    bwd[s] = b;
  }
  // Connect: accessible over kept arcs and coaccessible to a kept final
  std::vector<char> acc(n, 0), co(n, 0);
  acc[0] = 1;
  for (int32_t s = 0; s < n; s++) if (acc[s]) for (int64_t k = L.off[s]; k < L.off[s + 1]; k++) if (keep_arc[k]) acc[L.next[k]] = 1;
  for (int32_t s = n - 1; s >= 0; s--) {
    bool c = keep_fin[s];
    for (int64_t k = L.off[s]; k < L.off[s + 1] && !c; k++) if (keep_arc[k] && co[L.next[k]]) c = true;
    co[s] = c;
  }
  std::vector<int32_t> id(n, -1);
  int32_t m = 0;
  for (int32_t s = 0; s < n; s++) if (acc[s] && co[s]) id[s] = m++;
  arcs->clear();
  fin->assign(m, LW{kInfF, kInfF});
  for (int32_t s = 0; s < n; s++) {
    if (id[s] < 0) continue;
    if (keep_fin[s]) (*fin)[id[s]] = L.fin[s];
    for (int64_t k = L.off[s]; k < L.off[s + 1]; k++)
      if (keep_arc[k] && id[L.next[k]] >= 0) arcs->push_back(RawArc{id[s], L.label[k], L.tid[k], id[L.next[k]], L.w[k]});
  }
  *n_out = m;
}

// Compact lattice under construction / transformation (state 0 = start)
struct CArc { int32_t label, next; LW w; std::vector<int32_t> str; };
struct CLat {
  std::vector<std::vector<CArc>> arcs;
  std::vector<char> is_final;
  std::vector<LW> fw;
  std::vector<std::vector<int32_t>> fs;
  int32_t n() const { return static_cast<int32_t>(arcs.size()); }
  void Resize(int32_t k) { arcs.assign(k, {}); is_final.assign(k, 0); fw.assign(k, LW{kInfF, kInfF}); fs.assign(k, {}); }
};

// DeterminizeLatticePruned :1202-1306 with its retry loop; emit(pass) converts the finished pass
template <class Emit>
bool DeterminizeWithRetry(Pass *P, const Lat *L, double beam, const Options &o, Emit emit) {
  Lat pruned;
  const Lat *cur = L;
  for (int iter = 0; iter < 10; iter++) {
    double effective_beam = beam;
    const bool ans = P->Run(*cur, beam, o, &effective_beam);
    if (effective_beam >= beam * o.retry_cutoff || beam == kInfD || iter + 1 == 10) {
      emit(*P);
      return ans;
    }
    if (effective_beam < 0.0) effective_beam = 0.0;
    double new_beam = beam * std::sqrt(effective_beam / beam);
    if (new_beam < 0.5 * beam) new_beam = 0.5 * beam;
    beam = new_beam;
    int32_t n2;
    std::vector<RawArc> arcs;
    std::vector<LW> fin;
    PruneToRaw(*cur, static_cast<float>(beam), &n2, &arcs, &fin);
    if (n2 == 0) { emit(*P); return ans; }
    Lat next;
    BuildLat(n2, 0, arcs, fin, &next);
    pruned = std::move(next);
    cur = &pruned;
  }
  return false;
}

// Output(MutableFst<CompactArc>) :64-120
void EmitCompact(Pass &P, CLat *C) {
  const int32_t n = static_cast<int32_t>(P.out.size());
  C->Resize(n);
  for (int32_t s = 0; s < n; s++)
    for (const TempArc &t : P.out_arcs[s]) {
      if (t.next == -1) { C->is_final[s] = 1; C->fw[s] = t.w; P.strs.ToVector(t.str, &C->fs[s]); }
      else { C->arcs[s].push_back(CArc{t.label, t.next, t.w, {}}); P.strs.ToVector(t.str, &C->arcs[s].back().str); }
    }
}
// Output(MutableFst<Arc>) :125-196 as raw arcs of a state-level lattice (extra states carry the strings)
void EmitStateLevel(Pass &P, int32_t *n_out, std::vector<RawArc> *arcs, std::vector<LW> *fin) {
  const int32_t n0 = static_cast<int32_t>(P.out.size());
  int32_t n = n0;
  arcs->clear();
  fin->assign(n, LW{kInfF, kInfF});
  std::vector<int32_t> seq;
  const LW one{0.f, 0.f};
  for (int32_t s = 0; s < n0; s++)
    for (const TempArc &t : P.out_arcs[s]) {
      P.strs.ToVector(t.str, &seq);
      if (t.next == -1) {
        int32_t cur = s;
        for (size_t i = 0; i < seq.size(); i++) {
          const int32_t nx = n++;
          fin->push_back(LW{kInfF, kInfF});
          arcs->push_back(RawArc{cur, 0, seq[i], nx, i == 0 ? t.w : one});
          cur = nx;
        }
        (*fin)[cur] = seq.empty() ? t.w : one;
      } else {
        int32_t cur = s;
        for (size_t i = 0; i + 1 < seq.size(); i++) {
          const int32_t nx = n++;
          fin->push_back(LW{kInfF, kInfF});
          arcs->push_back(RawArc{cur, i == 0 ? t.label : 0, seq[i], nx, i == 0 ? t.w : one});
          cur = nx;
        }
        arcs->push_back(RawArc{cur, seq.size() <= 1 ? t.label : 0, seq.empty() ? 0 : seq.back(), t.next, seq.size() <= 1 ? t.w : one});
      }
    }
  *n_out = n;
}

// topological renumbering of a compact lattice (state 0 = start stays first); false on a cycle
bool TopSortCompact(CLat *C) {
  const int32_t n = C->n();
  std::vector<int32_t> indeg(n, 0), order;
  for (int32_t s = 0; s < n; s++) for (const CArc &a : C->arcs[s]) indeg[a.next]++;
  if (n > 0 && indeg[0] == 0) order.push_back(0);
  for (int32_t s = 1; s < n; s++) if (indeg[s] == 0) order.push_back(s);
  for (size_t h = 0; h < order.size(); h++)
    for (const CArc &a : C->arcs[order[h]]) if (--indeg[a.next] == 0) order.push_back(a.next);
  if (static_cast<int32_t>(order.size()) != n) return false;
  std::vector<int32_t> id(n);
  for (int32_t i = 0; i < n; i++) id[order[i]] = i;
  CLat D;
  D.Resize(n);
  for (int32_t s = 0; s < n; s++) {
    D.arcs[id[s]] = std::move(C->arcs[s]);
    for (CArc &a : D.arcs[id[s]]) a.next = id[a.next];
    D.is_final[id[s]] = C->is_final[s]; D.fw[id[s]] = C->fw[s]; D.fs[id[s]] = std::move(C->fs[s]);
  }
  *C = std::move(D);
  return true;
}
// fst::Connect: states reachable from state 0 from which a final state is reachable, renumbered in order
void ConnectCompact(CLat *C) {
  const int32_t n = C->n();
  if (n == 0) return;
  std::vector<char> acc(n, 0), co(n, 0);
  std::vector<int32_t> stack(1, 0);
  acc[0] = 1;
  while (!stack.empty()) {
    const int32_t s = stack.back();
    stack.pop_back();
    for (const CArc &a : C->arcs[s]) if (!acc[a.next]) { acc[a.next] = 1; stack.push_back(a.next); }
  }
  std::vector<std::vector<int32_t>> rev(n);
  for (int32_t s = 0; s < n; s++) for (const CArc &a : C->arcs[s]) rev[a.next].push_back(s);
  for (int32_t s = 0; s < n; s++) if (C->is_final[s]) { co[s] = 1; stack.push_back(s); }
  while (!stack.empty()) {
    const int32_t s = stack.back();
    stack.pop_back();
    for (int32_t r : rev[s]) if (!co[r]) { co[r] = 1; stack.push_back(r); }
  }
  std::vector<int32_t> id(n, -1);
  int32_t m = 0;
  for (int32_t s = 0; s < n; s++) if (acc[s] && co[s]) id[s] = m++;
  CLat D;
  if (id[0] < 0) { *C = std::move(D); return; }
  D.Resize(m);
  for (int32_t s = 0; s < n; s++) {
    if (id[s] < 0) continue;
    for (CArc &a : C->arcs[s]) if (id[a.next] >= 0) { a.next = id[a.next]; D.arcs[id[s]].push_back(std::move(a)); }
    D.is_final[id[s]] = C->is_final[s]; D.fw[id[s]] = C->fw[s]; D.fs[id[s]] = std::move(C->fs[s]);
  }
  *C = std::move(D);
}

// ---------------------------------------------------------------- push-lattice.cc
// GetString :55-86: the first `len` transition-ids on a path from `state` (the first arc / the final string: "an arbitrary path")
void PathString(const CLat &C, int32_t state, int64_t arc_idx, int32_t *out, size_t len) {
  while (len > 0) {
    if (arc_idx == -1 && C.is_final[state]) { std::copy(C.fs[state].begin(), C.fs[state].begin() + len, out); return; }
    const CArc &a = C.arcs[state][arc_idx == -1 ? 0 : arc_idx];
    if (a.str.size() >= len) { std::copy(a.str.begin(), a.str.begin() + len, out); return; }
    std::copy(a.str.begin(), a.str.end(), out);
    out += a.str.size();
    len -= a.str.size();
    state = a.next;
    arc_idx = -1;
  }
}
bool PushStrings(CLat *C) {   // PushCompactLatticeStrings :30-205
  if (!TopSortCompact(C)) return false;
  const int32_t n = C->n();
  std::vector<int32_t> shift(n, 0);
  std::vector<int32_t> s1, s2;
  for (int32_t st = n - 1; st > 0; st--) {
    const size_t na = C->arcs[st].size();
    if (na == 0) { shift[st] = static_cast<int32_t>(C->fs[st].size()); continue; }
    int32_t sh = std::numeric_limits<int32_t>::max();
    if (C->is_final[st]) sh = std::min(sh, static_cast<int32_t>(C->fs[st].size()));
    for (const CArc &a : C->arcs[st]) sh = std::min(sh, shift[a.next] + static_cast<int32_t>(a.str.size()));
    // CheckForConflict :88-128
    if (na + (C->is_final[st] ? 1 : 0) > 1 && sh > 0) {
      s1.resize(sh); s2.resize(sh);
      size_t arc;
      if (C->is_final[st]) { std::copy(C->fs[st].begin(), C->fs[st].begin() + sh, s1.begin()); arc = 0; }
      else { PathString(*C, st, 0, s1.data(), s1.size()); arc = 1; }
      for (; arc < na; arc++) {
        PathString(*C, st, static_cast<int64_t>(arc), s2.data(), s2.size());
        size_t k = 0;
        while (k < s1.size() && s1[k] == s2[k]) k++;
        if (k != s1.size()) { sh = static_cast<int32_t>(k); s1.resize(sh); s2.resize(sh); }
      }
    }
    shift[st] = sh;
  }
  for (int32_t st = 0; st < n; st++) {   // ApplyShifts :165-199 (later states are still unshifted when they are read)
    const int32_t sh = shift[st];
    for (CArc &a : C->arcs[st]) {
      const size_t orig = a.str.size(), nsh = shift[a.next];
      std::vector<int32_t> s(a.str);
      s.resize(orig + nsh);
      PathString(*C, a.next, -1, s.data() + orig, nsh);
      a.str.assign(s.begin() + sh, s.end());
    }
    if (C->is_final[st]) C->fs[st].erase(C->fs[st].begin(), C->fs[st].begin() + sh);
  }
  return true;
}
bool PushWeights(CLat *C) {   // PushCompactLatticeWeights :212-271
  if (!TopSortCompact(C)) return false;
  const int32_t n = C->n();
  if (n == 0) return true;
  std::vector<LW> to_end(n);
  for (int32_t s = n - 1; s >= 0; s--) {
    LW t = C->is_final[s] ? C->fw[s] : LW{kInfF, kInfF};
    for (const CArc &a : C->arcs[s]) {
      const LW c = Times(a.w, to_end[a.next]);
      if (CompareW(t, c) < 0) t = c;    // Plus(t, c) = (Compare(t, c) >= 0 ? t : c)
    }
    to_end[s] = t;
  }
  to_end[0] = LW{0.f, 0.f};
  for (int32_t s = 0; s < n; s++) {
    if (IsZero(to_end[s])) continue;
    for (CArc &a : C->arcs[s]) if (!IsZero(to_end[a.next])) a.w = Times(a.w, Divide(to_end[a.next], to_end[s]));
    if (C->is_final[s]) C->fw[s] = Divide(C->fw[s], to_end[s]);
  }
  return true;
}

// ---------------------------------------------------------------- minimize-lattice.cc
bool Minimize(CLat *C, float delta) {
  if (!TopSortCompact(C)) return false;
  const int32_t n = C->n();
  auto str_hash = [](const std::vector<int32_t> &v) {   // kaldi::VectorHasher, 0 -> 53281
    size_t h = 0;
    for (int32_t x : v) { h *= 7853; h += static_cast<size_t>(x); }
    return h == 0 ? static_cast<size_t>(53281) : h;
  };
  std::vector<size_t> hash(n);
  for (int32_t s = n - 1; s >= 0; s--) {   // ComputeStateHashValues :100-127
    size_t h = C->is_final[s] ? 607 * str_hash(C->fs[s]) : 33317;
    for (const CArc &a : C->arcs[s]) {
      size_t label = static_cast<size_t>(a.label);
      if (label == 0) label = 51907;
      h += 1447 * label * (1 + str_hash(a.str) * (a.next > s ? hash[a.next] : 1));
    }
    hash[s] = h;
  }
  std::vector<int32_t> map(n);
  for (int32_t s = 0; s < n; s++) map[s] = s;
  std::unordered_map<size_t, std::vector<int32_t>> groups;
  for (int32_t s = 0; s < n; s++) groups[hash[s]].push_back(s);
  struct A { int32_t label, next; LW w; const std::vector<int32_t> *str; };
  std::vector<A> xs, ys;
  auto equivalent = [&](int32_t s, int32_t t) {   // Equivalent :152-197
    const LW fs = C->is_final[s] ? C->fw[s] : LW{kInfF, kInfF}, ft = C->is_final[t] ? C->fw[t] : LW{kInfF, kInfF};
    if (!(ApproxEqualW(fs, ft, delta) && C->fs[s] == C->fs[t])) return false;
    if (C->arcs[s].size() != C->arcs[t].size()) return false;
    for (int it = 0; it < 2; it++) {
      std::vector<A> &v = it == 0 ? xs : ys;
      v.clear();
      for (const CArc &a : C->arcs[it == 0 ? s : t]) v.push_back(A{a.label, map[a.next], a.w, &a.str});
      std::sort(v.begin(), v.end(), [](const A &p, const A &q) { return p.label != q.label ? p.label < q.label : p.next < q.next; });
    }
    for (size_t i = 0; i < xs.size(); i++)
      if (xs[i].next != ys[i].next || xs[i].label != ys[i].label || !ApproxEqualW(xs[i].w, ys[i].w, 0.0009765625f) || *xs[i].str != *ys[i].str) return false;
    return true;
  };
  int32_t removed = 0;
  for (int32_t s = n - 1; s >= 0; s--)   // ComputeStateMap :199-247
    for (int32_t t : groups[hash[s]])
      if (t > s && map[t] == t && equivalent(s, t)) { map[s] = t; removed++; break; }
  if (removed == 0) return true;
  // ModifyModel :249-277
  for (int32_t s = 0; s < n; s++) {
    if (map[s] != s) { C->arcs[s].clear(); C->is_final[s] = 0; continue; }
    for (CArc &a : C->arcs[s]) a.next = map[a.next];
  }
  ConnectCompact(C);
  return true;
}

// per-thread workspace
struct Workspace {
  Pass pass;
  Lat lat;
};
Workspace &Tls() {
  static thread_local Workspace w;
  return w;
}

}  // namespace

extern "C" {

KhCompactLattice *kh_determinize_lattice_phone_pruned(int n_states, int n_arcs, const int32_t *arc_src, const int32_t *arc_dst,
                                                      const int32_t *arc_ilabel, const int32_t *arc_olabel, const float *arc_graph,
                                                      const float *arc_acoustic, const float *state_final, const int32_t *tid_phone,
                                                      int n_tid, double beam, float delta, int64_t max_mem, int phone_determinize,
                                                      int word_determinize, int minimize) {
  if (n_states < 0 || n_arcs < 0 || (n_arcs > 0 && (!arc_src || !arc_dst || !arc_ilabel || !arc_olabel || !arc_graph || !arc_acoustic)) ||
      (n_states > 0 && !state_final) || !(beam > 0.0) || (phone_determinize && (!tid_phone || n_tid <= 0))) {
    SetError("kh_determinize_lattice_phone_pruned: bad arguments");
    return nullptr;
  }
  Options o;
  o.delta = delta;
  o.max_mem = max_mem;
  if (const char *e = getenv("KH_DETERMINIZE_SHARE_MINIMAL")) o.share_minimal = atoi(e) != 0;
  // KH_DETERMINIZE_PROFILE=1: where a call's time goes (microseconds, to stderr)
  static const bool prof = getenv("KH_DETERMINIZE_PROFILE") != nullptr && atoi(getenv("KH_DETERMINIZE_PROFILE")) != 0;
  auto t_prev = std::chrono::steady_clock::now();
  double t_part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto lap = [&](int k) {
    if (!prof) return;
    const auto now = std::chrono::steady_clock::now();
    t_part[k] += std::chrono::duration<double, std::micro>(now - t_prev).count();
    t_prev = now;
  };
  Workspace &ws = Tls();
  // Invert + TopSort + ArcSort (:1504-1514): the pass's label is the word, its tid the transition-id
  std::vector<RawArc> arcs(n_arcs);
  for (int j = 0; j < n_arcs; j++) {
    if (arc_src[j] < 0 || arc_src[j] >= n_states || arc_dst[j] < 0 || arc_dst[j] >= n_states) {
      SetError("kh_determinize_lattice_phone_pruned: arc %d out of range", j);
      return nullptr;
    }
    arcs[j] = RawArc{arc_src[j], arc_olabel[j], arc_ilabel[j], arc_dst[j], LW{arc_graph[j], arc_acoustic[j]}};
  }
  std::vector<LW> fin(n_states);
  for (int s = 0; s < n_states; s++) fin[s] = state_final[s] != kInfF ? LW{state_final[s], 0.0f} : LW{kInfF, kInfF};
  KhCompactLattice *K = new KhCompactLattice();
  if (n_states == 0) return K;
  const char *cycle = "kh_determinize_lattice_phone_pruned: the lattice has a cycle (empty words in the lexicon, or epsilon cycles in the LM)";
  bool ans = true;
  CLat C;
  int32_t n_cur = n_states;
  if (!phone_determinize && !word_determinize) {
    // "copying lattice without determinization" :1421-1427
  } else if (phone_determinize) {
    // DeterminizeLatticeInsertPhones :1310-1360 (arcs out of the start state keep their labels)
    int32_t highest = 0;
    for (const RawArc &a : arcs) highest = std::max(highest, a.label);
    const int32_t first_phone_label = highest + 1;
    // (the wrapper has ArcSorted on the word by now, :1513-1514, and the phones go into the arcs in place: the depth-first
    // numbering of the first pass's TopSort follows a state's arcs in THAT order)
    std::stable_sort(arcs.begin(), arcs.end(), [](const RawArc &x, const RawArc &y) { return x.label < y.label; });
    const size_t m0 = arcs.size();
    for (size_t j = 0; j < m0; j++) {
      RawArc a = arcs[j];
      if (a.src == 0 || a.tid == 0 || a.tid >= n_tid || tid_phone[a.tid] == 0) continue;
      const int32_t ph = first_phone_label + tid_phone[a.tid];
      if (a.label == 0) {
        a.label = ph;
      } else {
        const int32_t extra = n_cur++;
        fin.push_back(LW{kInfF, kInfF});
        arcs.push_back(RawArc{extra, ph, 0, a.next, LW{0.f, 0.f}});
        a.next = extra;
      }
      arcs[j] = a;
    }
    lap(0);
    if (!BuildLat(n_cur, 0, arcs, fin, &ws.lat, true)) { SetError("%s", cycle); delete K; return nullptr; }   // TopSort :1410
    lap(1);
    // first pass -> state-level lattice, phones deleted (:1386-1404)
    std::vector<RawArc> arcs2;
    std::vector<LW> fin2;
    int32_t n2 = 0;
    ans = DeterminizeWithRetry(&ws.pass, &ws.lat, beam, o, [&](Pass &P) { EmitStateLevel(P, &n2, &arcs2, &fin2); }) && ans;
    lap(2);
    for (RawArc &a : arcs2) if (a.label >= first_phone_label) a.label = 0;
    arcs.swap(arcs2);
    fin.swap(fin2);
    n_cur = n2;
  }
  lap(0);
  // (after a first pass: TopSort :1417, unconditional; else the wrapper's own, which keeps a sorted lattice's numbering)
  if (n_cur > 0 && !BuildLat(n_cur, 0, arcs, fin, &ws.lat, phone_determinize != 0)) { SetError("%s", cycle); delete K; return nullptr; }
  lap(3);
  if (n_cur == 0) {
    // (an empty first pass)
  } else if (word_determinize) {
    ans = DeterminizeWithRetry(&ws.pass, &ws.lat, beam, o, [&](Pass &P) { EmitCompact(P, &C); }) && ans;
    if (minimize) {
      ans = PushStrings(&C) && ans;
      ans = PushWeights(&C) && ans;
      ans = Minimize(&C, 0.0009765625f) && ans;
    }
  } else {
    // ConvertLattice(ifst, ofst, false): every arc keeps its word and carries its transition-id as a string
    const Lat &L = ws.lat;
    C.Resize(L.n);
    for (int32_t s = 0; s < L.n; s++) {
      if (!IsZero(L.fin[s])) { C.is_final[s] = 1; C.fw[s] = L.fin[s]; }
      for (int64_t k = L.off[s]; k < L.off[s + 1]; k++) {
        C.arcs[s].push_back(CArc{L.label[k], L.next[k], L.w[k], {}});
        if (L.tid[k] != 0) C.arcs[s].back().str.push_back(L.tid[k]);
      }
    }
  }
  lap(4);
  ConnectCompact(&C);   // :1517
  K->complete = ans ? 1 : 0;
  K->n_states = C.n();
  K->final_g.assign(C.n(), kInfF);
  K->final_a.assign(C.n(), kInfF);
  K->final_str_off.assign(1, 0);
  K->arc_str_off.assign(1, 0);
  for (int32_t s = 0; s < C.n(); s++) {
    for (const CArc &a : C.arcs[s]) {
      K->arc_src.push_back(s); K->arc_dst.push_back(a.next); K->arc_label.push_back(a.label);
      K->arc_g.push_back(a.w.g); K->arc_a.push_back(a.w.a);
      K->strings.insert(K->strings.end(), a.str.begin(), a.str.end());
      K->arc_str_off.push_back(static_cast<int32_t>(K->strings.size()));
    }
    if (C.is_final[s]) {
      K->final_g[s] = C.fw[s].g;
      K->final_a[s] = C.fw[s].a;
      K->final_strings.insert(K->final_strings.end(), C.fs[s].begin(), C.fs[s].end());
    }
    K->final_str_off.push_back(static_cast<int32_t>(K->final_strings.size()));
  }
  lap(5);
  if (prof)
    fprintf(stderr, "[kh_determinize profile] %d states, %d arcs: input + phone insertion %.0f us, first BuildLat %.0f, phone pass %.0f, second "
            "BuildLat (%d states) %.0f, word pass (+ minimize) %.0f, connect + output %.0f\n", n_states, n_arcs, t_part[0], t_part[1], t_part[2],
            n_cur, t_part[3], t_part[4], t_part[5]);
  return K;
}

// the word-level pass alone (phone_determinize = false): what a caller without a TransitionModel can ask for
KhCompactLattice *kh_determinize_lattice_pruned(int n_states, int n_arcs, const int32_t *arc_src, const int32_t *arc_dst,
                                                const int32_t *arc_ilabel, const int32_t *arc_olabel, const float *arc_graph,
                                                const float *arc_acoustic, const float *state_final, double beam, float delta,
                                                int64_t max_mem) {
  return kh_determinize_lattice_phone_pruned(n_states, n_arcs, arc_src, arc_dst, arc_ilabel, arc_olabel, arc_graph, arc_acoustic,
                                             state_final, nullptr, 0, beam, delta, max_mem, 0, 1, 0);
}

int kh_compact_lattice_sizes(const KhCompactLattice *c, int32_t *n_states, int32_t *n_arcs, int32_t *n_arc_string_labels,
                             int32_t *n_final_string_labels, int32_t *complete) {
  KH_CHECK_ARG(c && n_states && n_arcs && n_arc_string_labels && n_final_string_labels && complete);
  *n_states = c->n_states;
  *n_arcs = static_cast<int32_t>(c->arc_src.size());
  *n_arc_string_labels = static_cast<int32_t>(c->strings.size());
  *n_final_string_labels = static_cast<int32_t>(c->final_strings.size());
  *complete = c->complete;
  return KH_OK;
}

int kh_compact_lattice_get(const KhCompactLattice *c, int32_t *arc_src, int32_t *arc_dst, int32_t *arc_label, float *arc_graph,
                           float *arc_acoustic, int32_t *arc_string_offsets, int32_t *arc_strings, float *final_graph,
                           float *final_acoustic, int32_t *final_string_offsets, int32_t *final_strings) {
  KH_CHECK_ARG(c);
  const size_t m = c->arc_src.size(), n = c->n_states;
  if (arc_src) memcpy(arc_src, c->arc_src.data(), 4 * m);
  if (arc_dst) memcpy(arc_dst, c->arc_dst.data(), 4 * m);
  if (arc_label) memcpy(arc_label, c->arc_label.data(), 4 * m);
  if (arc_graph) memcpy(arc_graph, c->arc_g.data(), 4 * m);
  if (arc_acoustic) memcpy(arc_acoustic, c->arc_a.data(), 4 * m);
  if (arc_string_offsets) memcpy(arc_string_offsets, c->arc_str_off.data(), 4 * c->arc_str_off.size());
  if (arc_strings) memcpy(arc_strings, c->strings.data(), 4 * c->strings.size());
  if (final_graph) memcpy(final_graph, c->final_g.data(), 4 * n);
  if (final_acoustic) memcpy(final_acoustic, c->final_a.data(), 4 * n);
  if (final_string_offsets) memcpy(final_string_offsets, c->final_str_off.data(), 4 * c->final_str_off.size());
  if (final_strings) memcpy(final_strings, c->final_strings.data(), 4 * c->final_strings.size());
  return KH_OK;
}

void kh_compact_lattice_free(KhCompactLattice *c) { delete c; }

}  // extern "C"
