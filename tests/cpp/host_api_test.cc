// host_api_test.cc — exercises the C++ host layer (old-kaldi-git_amd/host/kaldi-hip.h)
// the way the reference's cudamatrix/cu-matrix-test.cc exercises CuMatrix<BaseFloat>:
// random shapes, CPU loops as the check (UnitTestCuMatrixAddMatMat :1038-1066,
// UnitTestCuSoftmax :1559-1587, UnitTestCuMatrixCopyRows :379-402 with -1 indices,
// UnitTestCuMatrixGroupPnorm :246, UnitTestCuMatrixSumColumnRanges :442,
// UnitTestCuMatrixLookup :2011, cu-math-test.cc Splice), plus a tiny decode.
// Needs a GPU; built by __graft_entry__.build(), run by tests/test_gpu_cpp_host.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../../old-kaldi-git_amd/host/kaldi-hip.h"

using namespace kaldi;

static float Rnd() { return static_cast<float>(rand()) / RAND_MAX * 2.f - 1.f; }
static std::vector<float> RandMat(int r, int c, float s = 1.f) {
  std::vector<float> m(static_cast<size_t>(r) * c);
  for (size_t i = 0; i < m.size(); i++) m[i] = Rnd() * s;
  return m;
}
#define CHECK(cond) do { if (!(cond)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); exit(1); } } while (0)
static void Near(float a, float b, float rel, float abs_tol = 1e-6f) {
  if (!(std::fabs(a - b) <= abs_tol + rel * std::fabs(b))) { printf("FAIL %g vs %g\n", a, b); exit(1); }
}

static void TestAddMatMat() {
  const int m = 200, k = 100, n = 190;
  std::vector<float> A = RandMat(m, k), B = RandMat(n, k), C0 = RandMat(m, n);
  CuMatrix<BaseFloat> a, b, c;
  a.CopyFromMat(A.data(), m, k, k); b.CopyFromMat(B.data(), n, k, k); c.CopyFromMat(C0.data(), m, n, n);
  c.AddMatMat(0.5f, a, kNoTrans, b, kTrans, 0.25f);
  std::vector<float> out(m * n);
  c.CopyToMat(out.data(), n);
  for (int i = 0; i < m; i += 7)
    for (int j = 0; j < n; j += 5) {
      double s = 0;
      for (int p = 0; p < k; p++) s += static_cast<double>(A[i * k + p]) * B[j * k + p];
      Near(out[i * n + j], static_cast<float>(0.5 * s + 0.25 * C0[i * n + j]), 1e-4f, 1e-4f);
    }
  bool threw = false;
  try { c.AddMatMat(1.f, a, kNoTrans, b, kNoTrans, 0.f); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);  // dimension mismatch -> KALDI_ASSERT -> exception
}

// CuSubMatrix<BaseFloat> / CuSubVector<BaseFloat> (cu-matrix.h:620-644): the same operations on views must give the
// same numbers as on owning matrices holding copies of the blocks (as cu-matrix-test.cc does
// with its Range() cases).
static void TestSubMatrixViews() {
  const int R = 50, Cc = 70;
  std::vector<float> X = RandMat(R, Cc, 2.f), Y = RandMat(40, 64, 1.f);
  CuMatrix<BaseFloat> big, other;
  big.CopyFromMat(X.data(), R, Cc, Cc);
  other.CopyFromMat(Y.data(), 40, 64, 64);
  // view = rows [5, 35), cols [3, 43) of big; owning copy of the same block
  CuSubMatrix<BaseFloat> view(big, 5, 30, 3, 40);
  CHECK(view.NumRows() == 30 && view.NumCols() == 40 && view.Stride() == big.Stride());
  CuMatrix<BaseFloat> blockcopy(view);
  std::vector<float> a(30 * 40), b(30 * 40);
  view.CopyToMat(a.data(), 40);
  for (int i = 0; i < 30; i++)
    for (int j = 0; j < 40; j++) CHECK(a[i * 40 + j] == X[(i + 5) * Cc + j + 3]);
  // softmax into a view of another matrix vs into an owning matrix
  CuMatrix<BaseFloat> out1(30, 40), out2(60, 90);
  CuSubMatrix<BaseFloat> out2v = out2.Range(7, 30, 11, 40);
  out1.ApplySoftMaxPerRow(blockcopy);
  out2v.ApplySoftMaxPerRow(view);
  out1.CopyToMat(a.data(), 40);
  out2v.CopyToMat(b.data(), 40);
  for (size_t i = 0; i < a.size(); i++) CHECK(a[i] == b[i]);
  // the rest of out2 is untouched (still zero)
  std::vector<float> whole(60 * 90);
  out2.CopyToMat(whole.data(), 90);
  CHECK(whole[0] == 0.f && whole[6 * 90 + 11] == 0.f && whole[7 * 90 + 10] == 0.f && whole[37 * 90 + 11] == 0.f);
  // AddMatMat on views: C_view = A_view * B_view^T
  CuSubMatrix<BaseFloat> A = big.Range(0, 20, 10, 32), Bv = other.ColRange(16, 32).RowRange(4, 24);
  CuMatrix<BaseFloat> Ac, Bc, C1(20, 24);
  Ac = A;
  Bc = Bv;
  C1.AddMatMat(1.0f, Ac, kNoTrans, Bc, kTrans, 0.0f);
  CuMatrix<BaseFloat> C2big(33, 50);
  CuSubMatrix<BaseFloat> C2 = C2big.Range(13, 20, 26, 24);
  C2.AddMatMat(1.0f, A, kNoTrans, Bv, kTrans, 0.0f);
  std::vector<float> c1(20 * 24), c2(20 * 24);
  C1.CopyToMat(c1.data(), 24);
  C2.CopyToMat(c2.data(), 24);
  for (size_t i = 0; i < c1.size(); i++) CHECK(c1[i] == c2[i]);
  // CuSubVector<BaseFloat>: a row of a matrix as the bias of AddVecToRows; a range of a vector as a scale
  CuSubVector<BaseFloat> row(big, 9);
  CHECK(row.Dim() == Cc);
  CuMatrix<BaseFloat> M1(8, Cc), M2(8, Cc);
  std::vector<float> rowh(X.begin() + 9 * Cc, X.begin() + 10 * Cc);
  CuVector<BaseFloat> rowc(rowh);
  M1.AddVecToRows(1.0f, rowc);
  M2.AddVecToRows(1.0f, row);
  std::vector<float> m1(8 * Cc), m2(8 * Cc);
  M1.CopyToMat(m1.data(), Cc);
  M2.CopyToMat(m2.data(), Cc);
  for (size_t i = 0; i < m1.size(); i++) CHECK(m1[i] == m2[i] && m1[i] == X[9 * Cc + i % Cc]);
  CuSubVector<BaseFloat> part = rowc.Range(10, 8);
  CuMatrix<BaseFloat> M3(8, 5);
  std::vector<float> ones(40, 1.f);
  M3.CopyFromMat(ones.data(), 8, 5, 5);
  M3.MulRowsVec(part);
  std::vector<float> m3(40);
  M3.CopyToMat(m3.data(), 5);
  for (int i = 0; i < 8; i++) CHECK(m3[i * 5 + 2] == rowh[10 + i]);
  // out-of-range views are assertion failures, as in the reference
  bool threw = false;
  try { CuSubMatrix<BaseFloat> bad(big, 40, 20, 0, 10); } catch (const std::exception &) { threw = true; }
  CHECK(threw);
}

static void TestSoftmaxPnormCopyRows() {
  const int r = 37, c = 60;
  std::vector<float> X = RandMat(r, c, 5.f);
  CuMatrix<BaseFloat> x, y(r, c);
  x.CopyFromMat(X.data(), r, c, c);
  y.ApplySoftMaxPerRow(x);
  std::vector<float> out(r * c);
  y.CopyToMat(out.data(), c);
  for (int i = 0; i < r; i++) {
    double mx = -1e30, s = 0;
    for (int j = 0; j < c; j++) mx = std::max<double>(mx, X[i * c + j]);
    for (int j = 0; j < c; j++) s += std::exp(X[i * c + j] - mx);
    for (int j = 0; j < c; j++) Near(out[i * c + j], static_cast<float>(std::exp(X[i * c + j] - mx) / s), 1e-5f, 1e-9f);
  }
  CuMatrix<BaseFloat> p(r, c / 10);
  p.GroupPnorm(x, 2.0f);
  std::vector<float> po(r * (c / 10));
  p.CopyToMat(po.data(), c / 10);
  for (int i = 0; i < r; i++)
    for (int g = 0; g < c / 10; g++) {
      double s = 0;
      for (int j = 0; j < 10; j++) s += static_cast<double>(X[i * c + g * 10 + j]) * X[i * c + g * 10 + j];
      Near(po[i * (c / 10) + g], static_cast<float>(std::sqrt(s)), 1e-5f);
    }
  std::vector<int32> idx(50);
  for (int i = 0; i < 50; i++) idx[i] = (rand() % (r + 1)) - 1;
  CuMatrix<BaseFloat> d(50, c);
  d.CopyRows(x, idx);
  std::vector<float> dout(50 * c);
  d.CopyToMat(dout.data(), c);
  for (int i = 0; i < 50; i++)
    for (int j = 0; j < c; j++) CHECK(dout[i * c + j] == (idx[i] < 0 ? 0.f : X[idx[i] * c + j]));
  std::vector<int32> offs = {-3, 0, 2};
  CuMatrix<BaseFloat> sp(r, c * 3);
  cu::Splice(x, offs, &sp);
  std::vector<float> so(r * c * 3);
  sp.CopyToMat(so.data(), c * 3);
  for (int i = 0; i < r; i++)
    for (int o = 0; o < 3; o++) {
      int ri = std::min(std::max(i + offs[o], 0), r - 1);
      for (int j = 0; j < c; j++) CHECK(so[i * c * 3 + o * c + j] == X[ri * c + j]);
    }
  std::vector<int32> pairs = {0, 0, 3, 7, r - 1, c - 1};
  std::vector<float> lk;
  x.Lookup(pairs, &lk);
  CHECK(lk.size() == 3 && lk[0] == X[0] && lk[1] == X[3 * c + 7] && lk[2] == X[(r - 1) * c + c - 1]);
}

static void TestDecoder() {
  // 3-state left-to-right graph: 0 -(1:10/0.5)-> 1 -(2:0/0.25)-> 2(final 0.1), self loops tid 3 on 1
  std::vector<int64_t> off = {0, 1, 3, 3};
  std::vector<int32> il = {1, 3, 2}, ol = {10, 0, 0}, ns = {1, 1, 2};
  std::vector<float> w = {0.5f, 0.75f, 0.25f}, fin = {INFINITY, INFINITY, 0.1f};
  KhFst *fst = kh_fst_create(3, 0, off.data(), il.data(), ol.data(), w.data(), ns.data(), fin.data());
  CHECK(fst != NULL);
  LatticeFasterDecoderConfig cfg;
  LatticeFasterDecoder dec(fst, cfg, 1, 8);
  // 3 frames, 3 pdfs (tid - 1): choose loglikes so the path 1,3,2 wins
  std::vector<float> ll = {0.f, -5.f, -5.f, -5.f, -5.f, 0.f, -5.f, 0.f, -5.f};
  CuMatrix<BaseFloat> L;
  L.CopyFromMat(ll.data(), 3, 3, 3);
  std::vector<int32> offs = {0, 3};
  CHECK(dec.Decode(L.Data(), L.Stride(), offs, NULL));
  CHECK(dec.ReachedFinal(0));
  std::vector<int32> ali, words;
  float g, a;
  CHECK(dec.GetBestPath(0, &ali, &words, &g, &a));
  CHECK(ali.size() == 3 && ali[0] == 1 && ali[1] == 3 && ali[2] == 2);
  CHECK(words.size() == 1 && words[0] == 10);
  Near(g, 0.5f + 0.75f + 0.25f + 0.1f, 1e-6f);
  Near(a, 0.f, 1e-6f, 1e-6f);
  {  // the batched accessor gives the same path
    std::vector<int32> alis, wrds;
    std::vector<int64_t> aoff, woff;
    std::vector<float> gs, as;
    dec.GetBestPaths(0, 1, &alis, &aoff, &wrds, &woff, &gs, &as);
    CHECK(alis == ali && wrds == words && aoff.size() == 2 && aoff[1] == 3 && woff[1] == 1 && gs[0] == g && as[0] == a);
  }
  RawLattice lat;
  CHECK(dec.GetRawLattice(0, &lat));
  CHECK(lat.state_frame.size() == 4 && lat.arc_src.size() == 3);
  // the step behind the decoder (decoder-wrappers.cc:264-274): one word sequence -> one path whose
  // weights and transition-id strings add up to the best path's
  {
    CompactLatticeArrays clat;
    CHECK(DeterminizeLatticePruned(lat, 10.0, &clat));
    float cg = 0.f, ca = 0.f;
    std::vector<int32> cw, cali;
    int32 st = 0;
    for (int guard = 0; guard < 10; guard++) {
      int32 arc = -1;
      for (size_t i = 0; i < clat.arc_src.size(); i++)
        if (clat.arc_src[i] == st) { CHECK(arc < 0); arc = static_cast<int32>(i); }   // deterministic + a single path
      if (arc < 0) break;
      cg += clat.arc_graph[arc]; ca += clat.arc_acoustic[arc];
      if (clat.arc_label[arc] != 0) cw.push_back(clat.arc_label[arc]);
      for (int32 j = clat.arc_string_offsets[arc]; j < clat.arc_string_offsets[arc + 1]; j++) cali.push_back(clat.arc_strings[j]);
      st = clat.arc_dst[arc];
    }
    CHECK(clat.final_graph[st] < 1e30f);
    cg += clat.final_graph[st]; ca += clat.final_acoustic[st];
    for (int32 j = clat.final_string_offsets[st]; j < clat.final_string_offsets[st + 1]; j++) cali.push_back(clat.final_strings[j]);
    CHECK(cw == words && cali == ali);
    Near(cg, g, 1e-5f, 1e-5f);
    Near(ca, a, 1e-5f, 1e-5f);
  }
  // Decode(DecodableInterface *): a matrix-backed decodable with a TransitionIdToPdf map
  // (tid t -> pdf t - 1, what NULL meant above) gives the same result
  {
    std::vector<int32> t2p = {-1, 0, 1, 2};
    CuArray<int32> t2p_dev(t2p);
    DecodableMatrixMapped decodable(L, t2p_dev, t2p);
    CHECK(decodable.NumFramesReady() == 3 && decodable.NumIndices() == 3 && decodable.IsLastFrame(2) && !decodable.IsLastFrame(1));
    CHECK(decodable.LogLikelihood(1, 3) == 0.f && decodable.LogLikelihood(1, 1) == -5.f);
    DecodableInterface *itf = &decodable;
    CHECK(dec.Decode(itf));
    std::vector<int32> ali2, words2;
    float g2, a2;
    CHECK(dec.GetBestPath(0, &ali2, &words2, &g2, &a2));
    CHECK(ali2 == ali && words2 == words && g2 == g && a2 == a);
  }
  // the same utterance through the online call sequence, one frame + two frames
  LatticeFasterOnlineDecoder online(fst, cfg, 2, 8);
  std::vector<int32> s1(1, 1);
  online.InitDecoding(s1);
  std::vector<const BaseFloat *> rows(1, L.Data());
  std::vector<int32> nf(1, 1);
  online.AdvanceDecoding(s1, rows, L.Stride(), nf, NULL);
  CHECK(online.NumFramesDecoded(1) == 1);
  RawLattice partial;
  CHECK(online.GetRawLattice(1, &partial, false));
  CHECK(partial.state_frame.back() == 1 && partial.state_final.back() == 0.f);
  rows[0] = L.Data() + L.Stride();
  nf[0] = 2;
  online.AdvanceDecoding(s1, rows, L.Stride(), nf, NULL);
  online.FinalizeDecoding(s1);
  std::vector<int32> ali2, words2;
  float g2, a2;
  CHECK(online.GetBestPath(1, &ali2, &words2, &g2, &a2));
  CHECK(ali2 == ali && words2 == words && g2 == g && a2 == a);
  RawLattice lat2;
  CHECK(online.GetRawLattice(1, &lat2));
  CHECK(lat2.state_frame == lat.state_frame && lat2.arc_src == lat.arc_src && lat2.arc_graph == lat.arc_graph);
  kh_fst_destroy(fst);
}

static void TestNnetGmmLattice() {
  // 2-layer net: affine 3 -> 4, softmax; NnetComputation vs hand computation
  Nnet nnet;
  std::vector<float> W = {1, 0, 0,  0, 1, 0,  0, 0, 1,  1, 1, 1}, b = {0, 0, 0, -1};
  KhComponentDesc a;
  memset(&a, 0, sizeof(a));
  a.type = KH_AFFINE; a.input_dim = 3; a.output_dim = 4; a.linear = W.data(); a.bias = b.data();
  nnet.AddComponent(a);
  CHECK(nnet.InputDim() == 3 && nnet.OutputDim() == 4 && nnet.LeftContext() == 0);
  std::vector<float> x = {1, 2, 3,  0, 0, 0};
  CuMatrix<BaseFloat> X, Y;
  X.CopyFromMat(x.data(), 2, 3, 3);
  NnetComputation(nnet, X, true, &Y);
  std::vector<float> y(8);
  Y.CopyToMat(y.data(), 4);
  CHECK(y[0] == 1 && y[1] == 2 && y[2] == 3 && y[3] == 5 && y[7] == -1);
  // DiagGmm: one Gaussian N(0, I) in 2 dims: loglike(x) = -log(2 pi) - 0.5 |x|^2
  std::vector<float> w(1, 1.0f), mi(2, 0.0f), iv(2, 1.0f);
  DiagGmm gmm(w, mi, iv, 2);
  std::vector<float> d = {0, 0,  1, 2};
  CuMatrix<BaseFloat> D, LL;
  D.CopyFromMat(d.data(), 2, 2, 2);
  gmm.LogLikelihoods(D, &LL);
  std::vector<float> ll(2);
  LL.CopyToMat(ll.data(), 1);
  Near(ll[0], -1.8378770664f, 1e-5f);
  Near(ll[1], -1.8378770664f - 2.5f, 1e-5f);
  // lattice 0 -(1)-> 1 -(2)-> 3, 0 -(3)-> 2 -(2)-> 3: two paths with costs 1 and 2
  LatticeCsr lat;
  lat.arc_offsets = {0, 2, 3, 4, 4};
  lat.arc_ilabel = {1, 3, 2, 2};
  lat.arc_nextstate = {1, 2, 3, 3};
  lat.arc_graph = {0.5f, 1.0f, 0.5f, 1.0f};
  lat.arc_acoustic = {0, 0, 0, 0};
  const float inf = std::numeric_limits<float>::infinity();
  lat.state_final = {inf, inf, inf, 0.0f};
  std::vector<float> post;
  std::vector<int32> times;
  double tot = LatticeForwardBackward(lat, &post, NULL, &times);
  const double p1 = std::exp(-1.0), p2 = std::exp(-2.0);
  Near(static_cast<float>(tot), static_cast<float>(std::log(p1 + p2)), 1e-6f);
  Near(post[0], static_cast<float>(p1 / (p1 + p2)), 1e-6f);
  CHECK(times[3] == 2);
  // sMBR with the reference alignment {1, 2}: path 1 has accuracy 2, path 2 accuracy 1
  std::vector<int32> t2ph = {0, 1, 1, 1}, t2pdf = {0, 0, 1, 2}, sil, ali = {1, 2};
  std::vector<float> spost;
  double score = LatticeForwardBackwardMpeVariants(t2ph, t2pdf, sil, lat, ali, "smbr", false, &spost);
  Near(static_cast<float>(score), static_cast<float>((2 * p1 + 1 * p2) / (p1 + p2)), 1e-6f);
  // CompObjfAndDeriv
  CuMatrix<BaseFloat> out, deriv(1, 2);
  std::vector<float> o = {0.25f, 0.75f};
  out.CopyFromMat(o.data(), 1, 2, 2);
  std::vector<MatrixElement<BaseFloat>> lab(1);
  lab[0].row = 0; lab[0].column = 1; lab[0].weight = 2.0f;
  float objf, wt;
  deriv.CompObjfAndDeriv(lab, out, &objf, &wt);
  Near(objf, 2.0f * std::log(0.75f), 1e-6f);
  Near(wt, 2.0f, 1e-6f);
  std::vector<float> dv(2);
  deriv.CopyToMat(dv.data(), 2);
  Near(dv[1], 2.0f / 0.75f, 1e-6f);
  CHECK(CuDevice::Instantiate().ActiveGpuId() >= 0 && CuDevice::Instantiate().DoublePrecisionSupported());
  CuDevice::Instantiate().CheckGpuHealth();  // device GEMM against the host GEMM, < 1 % (cu-device.cc:509-527)
  int64_t fr = 0, to = 0;
  CuDevice::Instantiate().GetFreeMemory(&fr, &to);
  CHECK(fr > 0 && to >= fr);
}

// argv[1]: file written by tests/test_gpu_cpp_host.py: int32 n_samples, rows, cols; float wave[n];
// float mfcc[rows*cols] (api.Mfcc, hires options); float cmvn_deltas[rows * 3 cols] (CMVN then deltas)
static void TestFeatures(const char *path) {
  FILE *f = fopen(path, "rb");
  CHECK(f != NULL);
  int32 hdr[3];
  CHECK(fread(hdr, 4, 3, f) == 3);
  const int n = hdr[0], rows = hdr[1], cols = hdr[2];
  std::vector<float> wave(n), want(static_cast<size_t>(rows) * cols), want2(static_cast<size_t>(rows) * cols * 3);
  CHECK(fread(wave.data(), 4, n, f) == static_cast<size_t>(n));
  CHECK(fread(want.data(), 4, want.size(), f) == want.size());
  CHECK(fread(want2.data(), 4, want2.size(), f) == want2.size());
  fclose(f);
  MfccOptions opts;
  opts.num_bins = 40; opts.num_ceps = 40; opts.low_freq = 40; opts.high_freq = -200;
  Mfcc mfcc(opts);
  CuVector<BaseFloat> w(wave);
  CuMatrix<BaseFloat> feats;
  mfcc.Compute(w.Data(), n, &feats);
  CHECK(feats.NumRows() == rows && feats.NumCols() == cols);
  std::vector<float> got(want.size());
  feats.CopyToMat(got.data(), cols);
  for (size_t i = 0; i < got.size(); i++) Near(got[i], want[i], 1e-4f, 2e-4f);  // tables built in C++ vs numpy
  std::vector<double> stats;
  AccCmvnStats(feats, &stats);
  ApplyCmvn(stats, true, &feats);
  CuMatrix<BaseFloat> d;
  ComputeDeltas(2, 2, feats, &d);
  CHECK(d.NumCols() == 3 * cols);
  std::vector<float> got2(want2.size());
  d.CopyToMat(got2.data(), 3 * cols);
  for (size_t i = 0; i < got2.size(); i++) Near(got2[i], want2[i], 1e-3f, 1e-3f);
  // use_energy / raw_energy: C0 := log energy of the frame after DC removal (feature-mfcc.cc:138, :167-171),
  // the other columns unchanged; snip_edges = false: round(n / shift) frames
  {
    MfccOptions eo = opts;
    eo.use_energy = true;
    Mfcc emf(eo);
    CuMatrix<BaseFloat> ef;
    emf.Compute(w.Data(), n, &ef);
    CHECK(ef.NumRows() == rows && ef.NumCols() == cols);
    std::vector<float> e(want.size());
    ef.CopyToMat(e.data(), cols);
    for (int r = 0; r < rows; r += 37) {
      const int len = 400, shift = 160;
      double mean = 0.0, en = 0.0;
      for (int i = 0; i < len; i++) mean += wave[r * shift + i];
      const float c = -static_cast<float>(mean) / len;
      for (int i = 0; i < len; i++) { const float v = wave[r * shift + i] + c; en += static_cast<double>(v * v); }
      Near(e[static_cast<size_t>(r) * cols], logf(static_cast<float>(en)), 1e-4f, 1e-4f);
      for (int j = 1; j < cols; j++) CHECK(e[static_cast<size_t>(r) * cols + j] == got[static_cast<size_t>(r) * cols + j]);
    }
    MfccOptions so = opts;
    so.snip_edges = false;
    Mfcc smf(so);
    CuMatrix<BaseFloat> sf;
    smf.Compute(w.Data(), n, &sf);
    CHECK(sf.NumRows() == static_cast<int32>(n * 1.0f / 160 + 0.5f));
  }
}

// OnlineIvectorExtractor on a toy model: the rows of one period share one estimate, the estimate moves
// away from the prior as frames accumulate, utterances do not see each other
static void TestIvector() {
  KhIvectorConfig c;
  memset(&c, 0, sizeof(c));
  c.base_dim = 4; c.splice_left = 0; c.splice_right = 0; c.feat_dim = 4; c.num_gauss = 2; c.ivector_dim = 3; c.lda_cols = 4;
  c.cmn_window = 600; c.speaker_frames = 600; c.global_frames = 200; c.normalize_mean = 1; c.normalize_variance = 0;
  c.ivector_period = 2; c.num_gselect = 2; c.num_cg_iters = 15; c.min_post = 0.025f; c.posterior_scale = 0.5f; c.max_count = 0.f;
  c.prior_offset = 4.0;
  std::vector<float> lda(16, 0.f);
  for (int i = 0; i < 4; i++) lda[i * 4 + i] = 1.f;
  std::vector<double> gstats(10, 0.0);
  gstats[4] = 100.0;                                       // count; zero mean
  for (int i = 0; i < 4; i++) gstats[5 + i] = 100.0;      // unit variance
  // two unit-variance Gaussians at +-1: gconsts = log w - 0.5 (D log 2 pi + sum mu^2)
  std::vector<float> mi = {1.f, 1.f, 1.f, 1.f, -1.f, -1.f, -1.f, -1.f}, iv(8, 1.f), gc(2);
  for (int i = 0; i < 2; i++) gc[i] = logf(0.5f) - 0.5f * (4 * logf(6.2831853f) + 4.f);
  std::vector<double> M(2 * 4 * 3, 0.0), Si(2 * 16, 0.0);
  for (int g = 0; g < 2; g++)
    for (int d = 0; d < 4; d++) {
      M[(g * 4 + d) * 3 + 0] = (g == 0 ? 1.0 : -1.0) / c.prior_offset;   // mean_g = M_g [prior_offset, 0, 0]
      M[(g * 4 + d) * 3 + 1 + (d & 1)] = 0.5;
      Si[g * 16 + d * 4 + d] = 1.0;
    }
  OnlineIvectorExtractor ext(c, lda, gstats, gc, mi, iv, M, Si);
  CHECK(ext.IvectorDim() == 3);
  std::vector<float> x(7 * 4);
  for (size_t i = 0; i < x.size(); i++) x[i] = 0.3f * static_cast<float>((i * 7) % 5) - 0.2f;
  CuMatrix<BaseFloat> feats;
  feats.CopyFromMat(x.data(), 7, 4, 4);
  std::vector<int32> off = {0, 5, 7};
  CuMatrix<BaseFloat> iv_out;
  ext.Extract(feats, off, &iv_out);
  CHECK(iv_out.NumRows() == 7 && iv_out.NumCols() == 3);
  std::vector<float> h(21);
  iv_out.CopyToMat(h.data(), 3);
  for (int j = 0; j < 3; j++) {
    CHECK(h[0 * 3 + j] == h[1 * 3 + j] && h[2 * 3 + j] == h[3 * 3 + j] && h[5 * 3 + j] == h[6 * 3 + j]);
    CHECK(std::isfinite(h[4 * 3 + j]));
  }
  CHECK(fabsf(h[4 * 3 + 1]) + fabsf(h[4 * 3 + 2]) > 1e-3f);
  // the second utterance alone gives the same rows
  CuMatrix<BaseFloat> f2, o2;
  f2.CopyFromMat(x.data() + 5 * 4, 2, 4, 4);
  std::vector<int32> off2 = {0, 2};
  ext.Extract(f2, off2, &o2);
  std::vector<float> h2(6);
  o2.CopyToMat(h2.data(), 3);
  for (int j = 0; j < 6; j++) CHECK(h2[j] == h[15 + j]);
  // adaptation state: from fresh states the rows are those of Extract without a state; the second
  // utterance from the state the first one left sees its frames (a different estimate), and LimitFrames
  // scales the statistics down to max_remembered_frames * posterior_scale
  std::vector<double> st = ext.FreshStates(2);
  const int sd = ext.StateDim(), lo = 2 * 5 + 2;
  CHECK(sd == lo + 3 + 2);
  CuMatrix<BaseFloat> o3;
  ext.Extract(feats, off, &st, &o3);
  std::vector<float> h3(21);
  o3.CopyToMat(h3.data(), 3);
  for (int j = 0; j < 21; j++) CHECK(fabsf(h3[j] - h[j]) < 1e-6f);
  CHECK(fabs(st[4] - 5.0) < 1e-9 && fabs(st[sd + 4] - 2.0) < 1e-9);                   // CMVN counts
  CHECK(fabs(st[lo - 2] - 2.5) < 1e-6 && fabs(st[sd + lo - 2] - 1.0) < 1e-6);         // posterior_scale * frames
  std::vector<double> carried(st.begin(), st.begin() + sd);
  ext.LimitFrames(carried.data(), 3.f);
  CHECK(fabs(carried[4] - 3.0) < 1e-6 && fabs(carried[lo - 2] - 1.5) < 1e-6);
  CuMatrix<BaseFloat> o4;
  ext.Extract(f2, off2, &carried, &o4);
  std::vector<float> h4(6);
  o4.CopyToMat(h4.data(), 3);
  float moved = 0.f;
  for (int j = 0; j < 6; j++) moved += fabsf(h4[j] - h2[j]);
  CHECK(moved > 1e-3f && fabs(carried[lo - 2] - 2.5) < 1e-6);
}

int main(int argc, char **argv) {
  try {
    CuDevice::Instantiate().SelectGpuId("yes");
    printf("device: %s\n", CuDevice::Instantiate().DeviceGetName().c_str());
    TestAddMatMat();
    TestSoftmaxPnormCopyRows();
    TestSubMatrixViews();
    TestDecoder();
    TestNnetGmmLattice();
    TestIvector();
    if (argc > 1) TestFeatures(argv[1]);
  } catch (const std::exception &e) {
    printf("FAIL exception: %s\n", e.what());
    return 1;
  }
  printf("host_api_test: all tests passed\n");
  return 0;
}
