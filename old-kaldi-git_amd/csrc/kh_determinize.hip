// kh_determinize.hip — pruned lattice determinization on the host.
//
// Replaces DeterminizeLatticePhonePrunedWrapper (lat/determinize-lattice-pruned.cc:1497-1519,
// called from DecodeUtteranceLatticeFaster, decoder/decoder-wrappers.cc:264-274): the raw
// state-level lattice of the decoder -> a CompactLattice that is deterministic on WORDS, every
// word sequence keeping its best path (LatticeWeight order, fstext/lattice-weight.h:297-312,
// then the string order of :549-585) with that path's transition-id string, pruned to `beam`.
//
// Host code, as in the reference (it is the CPU tail behind the decoder; utterances are
// independent, so a batch is determinized on host threads).  Written from the algorithm's
// definition, not from the reference's 1500 lines: weighted subset construction in the
// semiring (LatticeWeight x transition-id string) — an output state is a set of
// (lattice state, residual weight, residual string); arcs without a word are followed inside
// the subset (the lattice is acyclic, so the closure is a relaxation to a fixed point); a
// subset is normalised by dividing out its best weight and its longest common string prefix,
// which go onto the arc; (state, label) expansions are taken from a priority queue in order
// of their best complete-path cost (forward cost + min over elements of residual + backward
// cost, determinize-lattice-pruned.cc:919-1001) and dropped beyond best + beam.
// The reference's optional first pass on phone + word labels (:1397-1421) is an efficiency
// device for very wide lattices (the language after the word-level pass is the same); only the
// word-level pass is built.  opts.minimize (default false) is not built.
//
// PARITY UNPINNED by the reference (src/lat needs OpenFst).  Pinned by
// tests/test_determinize.py: on small lattices against the enumeration of every path (the
// determinized lattice must hold exactly {word sequence -> best weight, its alignment} within
// the beam); on decoder lattices by the properties `lattice-equivalent` tests: deterministic,
// every sampled path's weight equals the best raw path of its word sequence, and its string is
// the alignment of a raw path with that weight.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <queue>
#include <unordered_map>
#include <vector>

#include "kh_common.h"

using namespace kh;

namespace {

const float kInfF = std::numeric_limits<float>::infinity();

struct LW { float g, a; };  // LatticeWeight (value1 = graph, value2 = acoustic)
inline LW Times(LW x, LW y) { return LW{x.g + y.g, x.a + y.a}; }
inline LW Divide(LW x, LW y) { return LW{x.g - y.g, x.a - y.a}; }
inline double Cost(LW x) { return static_cast<double>(x.g + x.a); }  // ConvertToCost lattice-weight.h:794-806
// fstext/lattice-weight.h:297-312: 1 if x is better (smaller) than y
inline int CompareW(LW x, LW y) {
  const float f1 = x.g + x.a, f2 = y.g + y.a;
  if (f1 < f2) return 1;
  if (f1 > f2) return -1;
  if (x.g < y.g) return 1;
  if (x.g > y.g) return -1;
  return 0;
}

// transition-id strings as nodes of a trie (LatticeStringRepository, determinize-lattice-pruned.cc)
struct Strings {
  struct Node { int32_t parent, label, depth; };
  std::vector<Node> nodes;                          // node 0 = the empty string
  std::unordered_map<uint64_t, int32_t> child;
  Strings() { nodes.push_back(Node{-1, 0, 0}); }
  int32_t Successor(int32_t s, int32_t label) {
    const uint64_t key = (static_cast<uint64_t>(static_cast<uint32_t>(s)) << 32) | static_cast<uint32_t>(label);
    auto it = child.find(key);
    if (it != child.end()) return it->second;
    const int32_t id = static_cast<int32_t>(nodes.size());
    nodes.push_back(Node{s, label, nodes[s].depth + 1});
    child.emplace(key, id);
    return id;
  }
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
  // the string s without its first `n` labels
  int32_t RemovePrefix(int32_t s, int n) {
    if (n == 0) return s;
    std::vector<int32_t> v;
    ToVector(s, &v);
    int32_t r = 0;
    for (size_t i = n; i < v.size(); i++) r = Successor(r, v[i]);
    return r;
  }
  int32_t Concatenate(int32_t a, int32_t b) {
    if (b == 0) return a;
    std::vector<int32_t> v;
    ToVector(b, &v);
    for (int32_t x : v) a = Successor(a, x);
    return a;
  }
  // determinize-lattice-pruned.cc:611-637: 1 if a is "better": shorter string, then the larger label sequence
  int Compare(int32_t a, int32_t b) const {
    if (a == b) return 0;
    if (nodes[a].depth > nodes[b].depth) return -1;
    if (nodes[a].depth < nodes[b].depth) return 1;
    // equal lengths: the first position where they differ is just below their common prefix
    int32_t la = 0, lb = 0;
    while (a != b) {
      la = nodes[a].label;
      lb = nodes[b].label;
      a = nodes[a].parent;
      b = nodes[b].parent;
    }
    return la < lb ? -1 : (la > lb ? 1 : 0);
  }
};

struct Elem { int32_t state; LW w; int32_t str; };

struct OutArc { int32_t label, next; LW w; int32_t str; };
struct OutState {
  std::vector<Elem> subset;  // minimal subset (sorted by state)
  std::vector<OutArc> arcs;
  double forward_cost = 0.0;
  bool is_final = false;
  LW final_w{0.f, 0.f};
  int32_t final_str = 0;
};

struct Task {
  int32_t state, label;
  double priority;
  std::vector<Elem> subset;
};
struct TaskWorse {
  bool operator()(const Task *x, const Task *y) const { return x->priority > y->priority; }
};

struct Determinizer {
  // input lattice (CSR by source state)
  int n = 0;
  std::vector<int64_t> off;
  const int32_t *dst, *il, *ol;  // il = transition-id, ol = word
  const float *g, *a, *fin;
  std::vector<int32_t> order;    // arcs sorted by source
  double beam = 0.0, cutoff = 0.0;
  float delta = 0.0009765625f;   // kDelta
  long long max_elems = 0, num_elems = 0;
  std::vector<double> backward;
  std::vector<char> has_word_arc;
  Strings strs;
  std::vector<OutState> out;
  std::map<std::vector<std::pair<int32_t, int32_t>>, std::vector<int32_t>> index;  // (state, string) list -> candidate output states
  std::priority_queue<Task *, std::vector<Task *>, TaskWorse> queue;

  int CompareElem(const LW &w1, int32_t s1, const LW &w2, int32_t s2) const {
    const int c = CompareW(w1, w2);
    return c != 0 ? c : strs.Compare(s1, s2);
  }

  // follow the arcs without a word inside the subset (determinize-lattice-pruned.cc:639-748).
  // States are expanded in topological order (a heap on the rank), so each is expanded once,
  // with its final best (weight, string).
  std::vector<int32_t> rank;        // topological rank of every lattice state
  std::vector<int32_t> slot_of;     // scratch: position of a state in the subset being closed, -1
  void EpsilonClosure(std::vector<Elem> *subset) {
    typedef std::pair<int32_t, int32_t> RS;  // (rank, state)
    std::priority_queue<RS, std::vector<RS>, std::greater<RS>> heap;
    for (size_t i = 0; i < subset->size(); i++) {
      slot_of[(*subset)[i].state] = static_cast<int32_t>(i);
      heap.push(RS(rank[(*subset)[i].state], (*subset)[i].state));
    }
    int32_t last = -1;
    while (!heap.empty()) {
      const int32_t s = heap.top().second;
      heap.pop();
      if (s == last) continue;  // (pushed once per improvement; expanded once: all improvements precede it)
      last = s;
      const Elem e = (*subset)[slot_of[s]];
      for (int64_t k = off[s]; k < off[s + 1]; k++) {
        const int32_t j = order[k];
        if (ol[j] != 0) continue;
        Elem ne;
        ne.state = dst[j];
        ne.w = Times(e.w, LW{g[j], a[j]});
        ne.str = il[j] != 0 ? strs.Successor(e.str, il[j]) : e.str;
        const int32_t at = slot_of[ne.state];
        if (at < 0) {
          slot_of[ne.state] = static_cast<int32_t>(subset->size());
          subset->push_back(ne);
          heap.push(RS(rank[ne.state], ne.state));
        } else if (CompareElem(ne.w, ne.str, (*subset)[at].w, (*subset)[at].str) == 1) {
          (*subset)[at] = ne;  // (its rank is larger than s's: not expanded yet, already in the heap)
        }
      }
    }
    for (const Elem &e : *subset) slot_of[e.state] = -1;
  }

  // divide out the best weight and the longest common string prefix (:793-824)
  void Normalize(std::vector<Elem> *subset, LW *tot, int32_t *common) {
    if (subset->empty()) { *tot = LW{0.f, 0.f}; *common = 0; return; }
    LW best = (*subset)[0].w;
    int32_t pre = (*subset)[0].str;
    for (size_t i = 1; i < subset->size(); i++) {
      if (CompareW((*subset)[i].w, best) == 1) best = (*subset)[i].w;
      pre = strs.CommonPrefix(pre, (*subset)[i].str);
    }
    const int n_pre = strs.nodes[pre].depth;
    for (Elem &e : *subset) {
      e.w = Divide(e.w, best);
      e.str = strs.RemovePrefix(e.str, n_pre);
    }
    *tot = best;
    *common = pre;
  }

  bool SameSubset(const std::vector<Elem> &x, const std::vector<Elem> &y) const {
    if (x.size() != y.size()) return false;
    for (size_t i = 0; i < x.size(); i++) {
      if (x[i].state != y[i].state || x[i].str != y[i].str) return false;
      const bool eq = (x[i].w.g == y[i].w.g && x[i].w.a == y[i].w.a) ||
                      std::fabs((x[i].w.g + x[i].w.a) - (y[i].w.g + y[i].w.a)) <= delta;  // ApproxEqual lattice-weight.h:359-364
      if (!eq) return false;
    }
    return true;
  }

  // output state of a subset (after closure, reduction to the states that matter, normalisation)
  int32_t StateOf(std::vector<Elem> *subset, double forward_cost, LW *tot, int32_t *common) {
    EpsilonClosure(subset);
    // ConvertToMinimal :508-525: keep the states that are final or have an arc with a word
    std::vector<Elem> minimal;
    for (const Elem &e : *subset)
      if (has_word_arc[e.state] || fin[e.state] != kInfF) minimal.push_back(e);
    std::sort(minimal.begin(), minimal.end(), [](const Elem &x, const Elem &y) { return x.state < y.state; });
    Normalize(&minimal, tot, common);
    std::vector<std::pair<int32_t, int32_t>> key;
    for (const Elem &e : minimal) key.emplace_back(e.state, e.str);
    std::vector<int32_t> &cands = index[key];
    for (int32_t c : cands)
      if (SameSubset(out[c].subset, minimal)) return c;
    const int32_t id = static_cast<int32_t>(out.size());
    cands.push_back(id);
    out.emplace_back();
    out[id].subset = minimal;
    out[id].forward_cost = forward_cost + Cost(*tot);
    num_elems += static_cast<long long>(minimal.size());
    ProcessFinal(id);
    ProcessTransitions(id);
    return id;
  }

  void ProcessFinal(int32_t id) {  // :750-791
    bool have = false;
    LW bw{0.f, 0.f};
    int32_t bs = 0;
    for (const Elem &e : out[id].subset) {
      if (fin[e.state] == kInfF) continue;
      const LW w = Times(e.w, LW{fin[e.state], 0.0f});
      if (!have || CompareElem(w, e.str, bw, bs) == 1) { bw = w; bs = e.str; have = true; }
    }
    out[id].is_final = have;
    out[id].final_w = bw;
    out[id].final_str = bs;
  }

  void ProcessTransitions(int32_t id) {  // :919-1001
    struct LE { int32_t label; Elem e; };
    std::vector<LE> all;
    for (const Elem &e : out[id].subset)
      for (int64_t k = off[e.state]; k < off[e.state + 1]; k++) {
        const int32_t j = order[k];
        if (ol[j] == 0) continue;
        Elem ne;
        ne.state = dst[j];
        ne.w = Times(e.w, LW{g[j], a[j]});
        ne.str = il[j] != 0 ? strs.Successor(e.str, il[j]) : e.str;
        all.push_back(LE{ol[j], ne});
      }
    std::sort(all.begin(), all.end(), [](const LE &x, const LE &y) {
      return x.label != y.label ? x.label < y.label : x.e.state < y.e.state;
    });
    size_t i = 0;
    while (i < all.size()) {
      Task *t = new Task;
      t->state = id;
      t->label = all[i].label;
      t->priority = std::numeric_limits<double>::infinity();
      while (i < all.size() && all[i].label == t->label) {
        const Elem &e = all[i].e;
        t->priority = std::min(t->priority, Cost(e.w) + backward[e.state]);
        // MakeSubsetUnique :826-861: one element per state, the best
        if (!t->subset.empty() && t->subset.back().state == e.state) {
          if (CompareElem(e.w, e.str, t->subset.back().w, t->subset.back().str) == 1) t->subset.back() = e;
        } else {
          t->subset.push_back(e);
        }
        i++;
      }
      t->priority += out[id].forward_cost;
      if (t->priority > cutoff) delete t;
      else queue.push(t);
    }
  }

  void ProcessTransition(Task *t) {  // :863-892
    LW tot, next_tot;
    int32_t common, next_common;
    Normalize(&t->subset, &tot, &common);
    const double forward_cost = out[t->state].forward_cost + Cost(tot);
    const int32_t next = StateOf(&t->subset, forward_cost, &next_tot, &next_common);
    OutArc arc;
    arc.label = t->label;
    arc.next = next;
    arc.w = Times(tot, next_tot);
    arc.str = strs.Concatenate(common, next_common);
    out[t->state].arcs.push_back(arc);
  }

  // returns false if it stopped early (memory limit)
  bool Run() {
    // backward costs (ComputeBackwardWeight :1030-1054) in reverse topological order
    std::vector<int32_t> indeg(n, 0), topo;
    for (int s = 0; s < n; s++)
      for (int64_t k = off[s]; k < off[s + 1]; k++) indeg[dst[order[k]]]++;
    for (int s = 0; s < n; s++) if (indeg[s] == 0) topo.push_back(s);
    for (size_t h = 0; h < topo.size(); h++) {
      const int s = topo[h];
      for (int64_t k = off[s]; k < off[s + 1]; k++)
        if (--indeg[dst[order[k]]] == 0) topo.push_back(dst[order[k]]);
    }
    if (static_cast<int>(topo.size()) != n) return false;  // cycle
    rank.assign(n, 0);
    for (int h = 0; h < n; h++) rank[topo[h]] = h;
    slot_of.assign(n, -1);
    backward.assign(n, std::numeric_limits<double>::infinity());
    has_word_arc.assign(n, 0);
    for (int h = n - 1; h >= 0; h--) {
      const int s = topo[h];
      double b = fin[s] != kInfF ? static_cast<double>(fin[s] + 0.0f) : std::numeric_limits<double>::infinity();
      for (int64_t k = off[s]; k < off[s + 1]; k++) {
        const int32_t j = order[k];
        b = std::min(b, static_cast<double>(g[j] + a[j]) + backward[dst[j]]);
        if (ol[j] != 0) has_word_arc[s] = 1;
      }
      backward[s] = b;
    }
    cutoff = backward[0] + beam;
    if (backward[0] == std::numeric_limits<double>::infinity()) return true;  // no complete path: empty output
    // InitializeDeterminization :1056-1109
    std::vector<Elem> start(1, Elem{0, LW{0.f, 0.f}, 0});
    LW tot;
    int32_t common;
    StateOf(&start, 0.0, &tot, &common);
    start_w = tot;
    start_str = common;
    // the start state carries its own normalisation on its arcs (there is no arc into it to put
    // it on; the reference leaves its start state un-normalised, :1075-1085): no later subset
    // may be merged with it
    index.clear();
    bool complete = true;
    while (!queue.empty()) {
      if (max_elems > 0 && num_elems > max_elems) { complete = false; break; }
      Task *t = queue.top();
      queue.pop();
      ProcessTransition(t);
      delete t;
    }
    while (!queue.empty()) { delete queue.top(); queue.pop(); }
    return complete;
  }
  LW start_w{0.f, 0.f};
  int32_t start_str = 0;
};

}  // namespace

struct KhCompactLattice {
  std::vector<int32_t> arc_src, arc_dst, arc_label, arc_str_off, strings;
  std::vector<float> arc_g, arc_a, final_g, final_a;
  std::vector<int32_t> final_str_off;  // [n_states + 1] into final_strings
  std::vector<int32_t> final_strings;
  int32_t n_states = 0;
  int complete = 1;
};

extern "C" {

KhCompactLattice *kh_determinize_lattice_pruned(int n_states, int n_arcs, const int32_t *arc_src, const int32_t *arc_dst,
                                                const int32_t *arc_ilabel, const int32_t *arc_olabel, const float *arc_graph,
                                                const float *arc_acoustic, const float *state_final, double beam, float delta,
                                                int max_mem) {
  if (n_states < 0 || n_arcs < 0 || (n_arcs > 0 && (!arc_src || !arc_dst || !arc_ilabel || !arc_olabel || !arc_graph || !arc_acoustic)) ||
      (n_states > 0 && !state_final) || !(beam > 0.0)) {
    SetError("kh_determinize_lattice_pruned: bad arguments");
    return nullptr;
  }
  Determinizer D;
  D.n = n_states;
  D.dst = arc_dst; D.il = arc_ilabel; D.ol = arc_olabel; D.g = arc_graph; D.a = arc_acoustic; D.fin = state_final;
  D.beam = beam;
  D.delta = delta;
  // max_mem is bytes in the reference (approximate: elements x sizeof(Element)); 0 = unlimited
  D.max_elems = max_mem > 0 ? std::max<long long>(64, max_mem / 32) : 0;
  D.off.assign(static_cast<size_t>(n_states) + 1, 0);
  for (int j = 0; j < n_arcs; j++) {
    if (arc_src[j] < 0 || arc_src[j] >= n_states || arc_dst[j] < 0 || arc_dst[j] >= n_states) {
      SetError("kh_determinize_lattice_pruned: arc %d out of range", j);
      return nullptr;
    }
    D.off[arc_src[j] + 1]++;
  }
  for (int s = 0; s < n_states; s++) D.off[s + 1] += D.off[s];
  D.order.resize(n_arcs);
  {
    std::vector<int64_t> fill(D.off.begin(), D.off.end() - 1);
    for (int j = 0; j < n_arcs; j++) D.order[fill[arc_src[j]]++] = j;
  }
  KhCompactLattice *C = new KhCompactLattice();
  if (n_states == 0) return C;
  C->complete = D.Run() ? 1 : 0;
  // Connect: keep the output states from which a final state is reachable (the start state is
  // reachable from itself; everything else was created by an arc from a reachable state)
  const int m = static_cast<int>(D.out.size());
  std::vector<char> co(m, 0);
  for (bool changed = true; changed;) {
    changed = false;
    for (int s = m - 1; s >= 0; s--) {
      if (co[s]) continue;
      bool ok = D.out[s].is_final;
      for (const OutArc &x : D.out[s].arcs) ok = ok || co[x.next];
      if (ok) { co[s] = 1; changed = true; }
    }
  }
  if (m == 0 || !co[0]) return C;  // empty lattice
  std::vector<int32_t> renum(m, -1);
  int32_t ns = 0;
  for (int s = 0; s < m; s++) if (co[s]) renum[s] = ns++;
  C->n_states = ns;
  C->final_g.assign(ns, kInfF);
  C->final_a.assign(ns, kInfF);
  C->final_str_off.assign(1, 0);
  C->arc_str_off.assign(1, 0);
  std::vector<int32_t> v;
  for (int s = 0; s < m; s++) {
    if (!co[s]) continue;
    const OutState &S = D.out[s];
    for (const OutArc &x : S.arcs) {
      if (!co[x.next]) continue;
      LW w = x.w;
      int32_t str = x.str;
      if (s == 0) {  // the start state's own normalisation goes onto its arcs (there is no arc into it)
        w = Times(D.start_w, w);
        str = D.strs.Concatenate(D.start_str, str);
      }
      C->arc_src.push_back(renum[s]);
      C->arc_dst.push_back(renum[x.next]);
      C->arc_label.push_back(x.label);
      C->arc_g.push_back(w.g);
      C->arc_a.push_back(w.a);
      D.strs.ToVector(str, &v);
      C->strings.insert(C->strings.end(), v.begin(), v.end());
      C->arc_str_off.push_back(static_cast<int32_t>(C->strings.size()));
    }
    if (S.is_final) {
      LW w = S.final_w;
      int32_t str = S.final_str;
      if (s == 0) {
        w = Times(D.start_w, w);
        str = D.strs.Concatenate(D.start_str, str);
      }
      C->final_g[renum[s]] = w.g;
      C->final_a[renum[s]] = w.a;
      D.strs.ToVector(str, &v);
      C->final_strings.insert(C->final_strings.end(), v.begin(), v.end());
    }
    C->final_str_off.push_back(static_cast<int32_t>(C->final_strings.size()));
  }
  return C;
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
