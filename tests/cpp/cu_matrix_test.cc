// cu_matrix_test.cc — device tests of the CuMatrix / CuVector primitives of the nnet2 forward path through
// old-kaldi-git_amd/host/kaldi-hip.h, written against the REFERENCE'S CLASS API (CuMatrix<Real>, CuVector<Real>,
// CuArray<T>, cu::Splice, MatrixElement<Real>, Int32Pair: the spellings and argument orders of cudamatrix/cu-matrix.h,
// cu-vector.h, cu-math.h), templated on Real and run for float AND double like the reference's CudaMatrixUnitTest<Real>().
// The test designs are this repo's own: every primitive is checked against algebraic identities and closed forms
// (A I = A, (A B)^T = B^T A^T, softmax rows sum to one and ignore a shift, a permutation and its inverse, the p-norm of a
// constant group, log(exp(x)) = x, ...) and against plain host loops over kaldi-matrix-lite.h's Matrix<Real>.
//
// plus, for this library: the same primitives on views (Range), element access / copies between precisions / Swap, what
// stays float-only on a <double> object, and LatticeFasterDecoder(fst, config) / Decode(&decodable) / GetRawLattice(&lat) as
// lattice-faster-decoder.h:101-140 declares them.  Runs on the GPU.  Prints "all tests passed" and exits 0, or the first
// failure and 1.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../old-kaldi-git_amd/host/kaldi-hip.h"

namespace kaldi {

// tolerance of a device result against a host loop in the same precision
template <typename Real> static Real Tol();
template <> float Tol<float>() { return 2e-4f; }
template <> double Tol<double>() { return 1e-11; }

#define EXPECT_NEAR(a, b, tol)                                                                                     \
  do {                                                                                                             \
    const double a_ = static_cast<double>(a), b_ = static_cast<double>(b);                                          \
    if (!(std::fabs(a_ - b_) <= static_cast<double>(tol) * (1.0 + std::fabs(a_) + std::fabs(b_)))) {                \
      char msg_[256];                                                                                              \
      snprintf(msg_, sizeof(msg_), "%s:%d: %s = %.17g, %s = %.17g", __FILE__, __LINE__, #a, a_, #b, b_);             \
      throw std::runtime_error(msg_);                                                                              \
    }                                                                                                              \
  } while (0)

// a matrix with a known pattern: value(r, c) = sin(0.37 r + 1.3 c + phase) * amp (no two rows alike)
template <typename Real>
static Matrix<Real> Pattern(MatrixIndexT rows, MatrixIndexT cols, double phase = 0.0, double amp = 1.0) {
  Matrix<Real> m(rows, cols);
  for (MatrixIndexT r = 0; r < rows; r++)
    for (MatrixIndexT c = 0; c < cols; c++) m(r, c) = static_cast<Real>(amp * std::sin(0.37 * r + 1.3 * c + phase));
  return m;
}

// ---- AddMatMat (cu-matrix.h: AddMatMat(alpha, A, transA, B, transB, beta)) ----------------------------------------
template <typename Real>
static void TestAddMatMat() {
  const MatrixIndexT m = 67, k = 129, n = 45;
  Matrix<Real> Ha = Pattern<Real>(m, k, 0.1), Hb = Pattern<Real>(k, n, 0.7);
  CuMatrix<Real> A(Ha), B(Hb);
  // (1) against the triple loop
  CuMatrix<Real> C(m, n);
  C.AddMatMat(1.0, A, kNoTrans, B, kNoTrans, 0.0);
  Matrix<Real> Hc(C);
  for (MatrixIndexT i = 0; i < m; i += 11)
    for (MatrixIndexT j = 0; j < n; j += 7) {
      double s = 0.0;
      for (MatrixIndexT q = 0; q < k; q++) s += static_cast<double>(Ha(i, q)) * Hb(q, j);
      EXPECT_NEAR(Hc(i, j), s, Tol<Real>() * 10);
    }
  // (2) (A B)^T = B^T A^T through the transpose flags
  CuMatrix<Real> Ct(n, m);
  Ct.AddMatMat(1.0, B, kTrans, A, kTrans, 0.0);
  Matrix<Real> Hct(Ct);
  for (MatrixIndexT i = 0; i < m; i++)
    for (MatrixIndexT j = 0; j < n; j++) EXPECT_NEAR(Hct(j, i), Hc(i, j), Tol<Real>());
  // (3) multiplying by the identity, from either side, with the other two flag combinations
  Matrix<Real> Hi(k, k);
  for (MatrixIndexT q = 0; q < k; q++) Hi(q, q) = 1.0;
  CuMatrix<Real> I(Hi), AI(m, k), IB(k, n);
  AI.AddMatMat(1.0, A, kNoTrans, I, kTrans, 0.0);
  Matrix<Real> Hai(AI);
  AssertEqual(Ha, Hai, 0.0);
  CuMatrix<Real> Bt(Hb, kTrans);   // n x k
  IB.AddMatMat(1.0, I, kTrans, Bt, kTrans, 0.0);
  Matrix<Real> Hib(IB);
  AssertEqual(Hb, Hib, 0.0);
  // (4) alpha and beta: C <- 2 A B - 1 C  = A B,  then  C <- -1 A B + 1 C = 0
  C.AddMatMat(2.0, A, kNoTrans, B, kNoTrans, -1.0);
  Matrix<Real> Hc2(C);
  for (MatrixIndexT i = 0; i < m; i++)
    for (MatrixIndexT j = 0; j < n; j++) EXPECT_NEAR(Hc2(i, j), Hc(i, j), Tol<Real>() * 10);
  C.AddMatMat(-1.0, A, kNoTrans, B, kNoTrans, 1.0);
  Matrix<Real> Hz(C);
  for (MatrixIndexT i = 0; i < m; i++)
    for (MatrixIndexT j = 0; j < n; j++) KALDI_ASSERT(std::fabs(Hz(i, j)) <= Tol<Real>() * 50);
}

// ---- ApplySoftMaxPerRow / ApplyLogSoftMaxPerRow -------------------------------------------------------------------
template <typename Real>
static void TestSoftmax() {
  const MatrixIndexT rows = 23;
  for (MatrixIndexT cols : {1, 9, 64, 65, 1500}) {
    Matrix<Real> Hx = Pattern<Real>(rows, cols, 0.3, 6.0);
    CuMatrix<Real> X(Hx), Y(rows, cols), L(rows, cols);
    Y.ApplySoftMaxPerRow(X);
    L.ApplyLogSoftMaxPerRow(X);
    Matrix<Real> Hy(Y), Hl(L);
    for (MatrixIndexT r = 0; r < rows; r++) {
      double mx = -1e300, sum = 0.0, tot = 0.0;
      for (MatrixIndexT c = 0; c < cols; c++) mx = std::max(mx, static_cast<double>(Hx(r, c)));
      for (MatrixIndexT c = 0; c < cols; c++) sum += std::exp(Hx(r, c) - mx);
      for (MatrixIndexT c = 0; c < cols; c++) {
        EXPECT_NEAR(Hy(r, c), std::exp(Hx(r, c) - mx) / sum, Tol<Real>());
        EXPECT_NEAR(Hl(r, c), Hx(r, c) - mx - std::log(sum), Tol<Real>());
        tot += Hy(r, c);
      }
      EXPECT_NEAR(tot, 1.0, Tol<Real>());
    }
    // a constant added to a row changes nothing
    Matrix<Real> Hs(Hx);
    for (MatrixIndexT r = 0; r < rows; r++)
      for (MatrixIndexT c = 0; c < cols; c++) Hs(r, c) += static_cast<Real>(3.0 + r);
    CuMatrix<Real> Xs(Hs), Ys(rows, cols);
    Ys.ApplySoftMaxPerRow(Xs);
    Matrix<Real> Hys(Ys);
    for (MatrixIndexT r = 0; r < rows; r++)
      for (MatrixIndexT c = 0; c < cols; c++) EXPECT_NEAR(Hys(r, c), Hy(r, c), Tol<Real>() * 20);
  }
}

// ---- CopyRows (std::vector<int32> of source rows, -1 = a zero row) -------------------------------------------------
template <typename Real>
static void TestCopyRows() {
  const MatrixIndexT rows = 37, cols = 21;
  Matrix<Real> Hm = Pattern<Real>(rows, cols, 0.9);
  CuMatrix<Real> M(Hm);
  // a permutation (stride 10 is coprime to 37) and its inverse give the matrix back
  std::vector<int32> perm(rows), inv(rows);
  for (int32 i = 0; i < rows; i++) { perm[i] = (i * 10 + 3) % rows; inv[perm[i]] = i; }
  CuMatrix<Real> P(rows, cols), Q(rows, cols);
  P.CopyRows(M, perm);
  Q.CopyRows(P, inv);
  Matrix<Real> Hp(P), Hq(Q);
  for (int32 i = 0; i < rows; i++)
    for (int32 j = 0; j < cols; j++) KALDI_ASSERT(Hp(i, j) == Hm(perm[i], j));
  AssertEqual(Hm, Hq, 0.0);
  // more destination rows than source rows, repeats, and -1
  std::vector<int32> idx = {-1, 0, 0, 36, -1, 5};
  CuMatrix<Real> R(6, cols);
  R.Set(9.0);   // (overwritten, also where the index is -1)
  R.CopyRows(M, idx);
  Matrix<Real> Hr(R);
  for (int32 i = 0; i < 6; i++)
    for (int32 j = 0; j < cols; j++) KALDI_ASSERT(Hr(i, j) == (idx[i] < 0 ? Real(0) : Hm(idx[i], j)));
}

// ---- GroupPnorm(src, power) ----------------------------------------------------------------------------------------
template <typename Real>
static void TestGroupPnorm() {
  const MatrixIndexT rows = 19;
  // a group of the constant c (> 0) repeated G times has p-norm c G^(1/p)
  for (int32 G : {1, 4, 10}) {
    const MatrixIndexT out_cols = 13;
    Matrix<Real> Hs(rows, out_cols * G);
    for (MatrixIndexT r = 0; r < rows; r++)
      for (MatrixIndexT c = 0; c < out_cols * G; c++) Hs(r, c) = static_cast<Real>((1 + r % 5) * 0.25 * ((c / G) % 2 ? -1 : 1));
    CuMatrix<Real> S(Hs), D(rows, out_cols);
    for (double p : {1.0, 2.0, 3.0, 0.5}) {
      D.GroupPnorm(S, static_cast<Real>(p));
      Matrix<Real> Hd(D);
      for (MatrixIndexT r = 0; r < rows; r++)
        for (MatrixIndexT c = 0; c < out_cols; c++) EXPECT_NEAR(Hd(r, c), (1 + r % 5) * 0.25 * std::pow(G, 1.0 / p), Tol<Real>() * 5);
    }
  }
  // general values, p = 2 and p = 1, against the loop; exact zeros in a group
  Matrix<Real> Hs = Pattern<Real>(rows, 70, 0.2, 2.0);
  for (MatrixIndexT c = 0; c < 7; c++) Hs(3, c) = 0.0;
  CuMatrix<Real> S(Hs), D2(rows, 10), D1(rows, 10);
  D2.GroupPnorm(S, 2.0);
  D1.GroupPnorm(S, 1.0);
  Matrix<Real> H2(D2), H1(D1);
  for (MatrixIndexT r = 0; r < rows; r++)
    for (MatrixIndexT c = 0; c < 10; c++) {
      double s2 = 0.0, s1 = 0.0;
      for (MatrixIndexT j = 0; j < 7; j++) { s2 += static_cast<double>(Hs(r, c * 7 + j)) * Hs(r, c * 7 + j); s1 += std::fabs(Hs(r, c * 7 + j)); }
      EXPECT_NEAR(H2(r, c), std::sqrt(s2), Tol<Real>());
      EXPECT_NEAR(H1(r, c), s1, Tol<Real>());
    }
  KALDI_ASSERT(H2(3, 0) == Real(0));
}

// ---- the element-wise set: ApplyLog, ApplyExp, ApplyPow, ApplyFloor, Scale ----------------------------------------
template <typename Real>
static void TestElementwise() {
  const MatrixIndexT rows = 31, cols = 77;
  Matrix<Real> Hx = Pattern<Real>(rows, cols, 0.5, 3.0);
  {  // log(exp(x)) = x
    CuMatrix<Real> X(Hx);
    X.ApplyExp();
    Matrix<Real> He(X);
    for (MatrixIndexT r = 0; r < rows; r += 3)
      for (MatrixIndexT c = 0; c < cols; c += 5) EXPECT_NEAR(He(r, c), std::exp(static_cast<double>(Hx(r, c))), Tol<Real>());
    X.ApplyLog();
    Matrix<Real> Hb(X);
    for (MatrixIndexT r = 0; r < rows; r++)
      for (MatrixIndexT c = 0; c < cols; c++) EXPECT_NEAR(Hb(r, c), Hx(r, c), Tol<Real>());
  }
  {  // Scale(a) then Scale(1 / a) for a power of two is exact; Scale(0) clears
    CuMatrix<Real> X(Hx);
    X.Scale(8.0);
    Matrix<Real> H8(X);
    for (MatrixIndexT r = 0; r < rows; r++)
      for (MatrixIndexT c = 0; c < cols; c++) KALDI_ASSERT(H8(r, c) == Hx(r, c) * Real(8));
    X.Scale(0.125);
    Matrix<Real> Hb(X);
    AssertEqual(Hx, Hb, 0.0);
    X.Scale(0.0);
    KALDI_ASSERT(X.Sum() == Real(0));
  }
  {  // ApplyFloor: nothing below the floor, everything above it untouched
    CuMatrix<Real> X(Hx);
    X.ApplyFloor(-0.75);
    Matrix<Real> Hf(X);
    for (MatrixIndexT r = 0; r < rows; r++)
      for (MatrixIndexT c = 0; c < cols; c++) KALDI_ASSERT(Hf(r, c) == (Hx(r, c) < Real(-0.75) ? Real(-0.75) : Hx(r, c)));
  }
  {  // ApplyPow on positives: squares, then square roots give the input back; a cube against pow()
    Matrix<Real> Hp(rows, cols);
    for (MatrixIndexT r = 0; r < rows; r++)
      for (MatrixIndexT c = 0; c < cols; c++) Hp(r, c) = static_cast<Real>(0.05 + std::fabs(Hx(r, c)));
    CuMatrix<Real> X(Hp);
    X.ApplyPow(2.0);
    Matrix<Real> Hs(X);
    for (MatrixIndexT r = 0; r < rows; r++)
      for (MatrixIndexT c = 0; c < cols; c++) EXPECT_NEAR(Hs(r, c), static_cast<double>(Hp(r, c)) * Hp(r, c), Tol<Real>());
    X.ApplyPow(0.5);
    Matrix<Real> Hb(X);
    for (MatrixIndexT r = 0; r < rows; r++)
      for (MatrixIndexT c = 0; c < cols; c++) EXPECT_NEAR(Hb(r, c), Hp(r, c), Tol<Real>());
    X.ApplyPow(3.0);
    Matrix<Real> Hc(X);
    for (MatrixIndexT r = 0; r < rows; r += 2)
      for (MatrixIndexT c = 0; c < cols; c += 3) EXPECT_NEAR(Hc(r, c), std::pow(static_cast<double>(Hb(r, c)), 3.0), Tol<Real>() * 5);
  }
}

// ---- rows and columns against vectors: CopyRowsFromVec, AddVecToRows, MulRowsVec, MulColsVec ----------------------
template <typename Real>
static void TestRowColumnVectors() {
  const MatrixIndexT rows = 29, cols = 53;
  Vector<Real> hr(rows), hc(cols);
  for (MatrixIndexT r = 0; r < rows; r++) hr(r) = static_cast<Real>(0.5 + 0.125 * r);
  for (MatrixIndexT c = 0; c < cols; c++) hc(c) = static_cast<Real>(-2.0 + 0.25 * c);
  CuVector<Real> vr(hr), vc(hc);
  CuMatrix<Real> M(rows, cols);
  M.CopyRowsFromVec(vc);                      // M(r, c) = g(c)
  M.MulRowsVec(vr);                           // f(r) g(c)
  Matrix<Real> H1(M);
  for (MatrixIndexT r = 0; r < rows; r++)
    for (MatrixIndexT c = 0; c < cols; c++) KALDI_ASSERT(H1(r, c) == hr(r) * hc(c));
  M.MulColsVec(vc);                           // f(r) g(c)^2
  M.AddVecToRows(2.0, vc, 0.5);               // 0.5 f(r) g(c)^2 + 2 g(c)
  Matrix<Real> H2(M);
  for (MatrixIndexT r = 0; r < rows; r++)
    for (MatrixIndexT c = 0; c < cols; c++)
      EXPECT_NEAR(H2(r, c), 0.5 * static_cast<double>(hr(r)) * hc(c) * hc(c) + 2.0 * hc(c), Tol<Real>());
  // CopyRowsFromVec with a vector that holds the WHOLE matrix row by row (cu-matrix.cc:1690-1705)
  Vector<Real> hall(rows * cols);
  for (MatrixIndexT i = 0; i < rows * cols; i++) hall(i) = static_cast<Real>(i % 97);
  CuVector<Real> vall(hall);
  M.CopyRowsFromVec(vall);
  Matrix<Real> H3(M);
  for (MatrixIndexT r = 0; r < rows; r++)
    for (MatrixIndexT c = 0; c < cols; c++) KALDI_ASSERT(H3(r, c) == static_cast<Real>((r * cols + c) % 97));
}

// ---- SumColumnRanges(src, CuArray<Int32Pair>) and Lookup(std::vector<Int32Pair>, std::vector<Real>*) ----------------
template <typename Real>
static void TestRangesAndLookup() {
  const MatrixIndexT rows = 17, cols = 60;
  Matrix<Real> Hs = Pattern<Real>(rows, cols, 1.1);
  CuMatrix<Real> S(Hs);
  // consecutive ranges of growing length that partition the columns (1 + 2 + ... + 10 = 55, then 5), and an empty one
  std::vector<Int32Pair> pr;
  int32 at = 0;
  for (int32 len = 1; len <= 10; len++) { Int32Pair p = {at, at + len}; pr.push_back(p); at += len; }
  Int32Pair last = {55, 60}, empty = {7, 7};
  pr.push_back(last);
  pr.push_back(empty);
  CuArray<Int32Pair> ranges(pr);
  CuMatrix<Real> D(rows, static_cast<MatrixIndexT>(pr.size()));
  D.SumColumnRanges(S, ranges);
  Matrix<Real> Hd(D);
  for (MatrixIndexT r = 0; r < rows; r++) {
    double row_total = 0.0, parts = 0.0;
    for (MatrixIndexT c = 0; c < cols; c++) row_total += Hs(r, c);
    for (size_t j = 0; j < pr.size(); j++) {
      double s = 0.0;
      for (int32 c = pr[j].first; c < pr[j].second; c++) s += Hs(r, c);
      EXPECT_NEAR(Hd(r, j), s, Tol<Real>());
      parts += Hd(r, j);
    }
    EXPECT_NEAR(parts, row_total, Tol<Real>() * 5);
    KALDI_ASSERT(Hd(r, pr.size() - 1) == Real(0));
  }
  // Lookup: the four corners, a diagonal walk, a repeat
  std::vector<Int32Pair> where;
  Int32Pair corners[4] = {{0, 0}, {0, cols - 1}, {rows - 1, 0}, {rows - 1, cols - 1}};
  for (int i = 0; i < 4; i++) where.push_back(corners[i]);
  for (int32 r = 0; r < rows; r++) { Int32Pair p = {r, (r * 7) % cols}; where.push_back(p); }
  where.push_back(corners[3]);
  std::vector<Real> got;
  S.Lookup(where, &got);
  KALDI_ASSERT(got.size() == where.size());
  for (size_t i = 0; i < where.size(); i++) KALDI_ASSERT(got[i] == Hs(where[i].first, where[i].second));
}

// ---- cu::Splice(src, CuArray<int32> frame_offsets, &tgt) -----------------------------------------------------------
template <typename Real>
static void TestSplice() {
  const MatrixIndexT rows = 40, cols = 6;
  Matrix<Real> Hs(rows, cols);
  for (MatrixIndexT r = 0; r < rows; r++)
    for (MatrixIndexT c = 0; c < cols; c++) Hs(r, c) = static_cast<Real>(100 * r + c);   // the row is readable from the value
  CuMatrix<Real> S(Hs);
  std::vector<int32> off = {-3, 0, 0, 2, 45, -45};
  CuArray<int32> offsets(off);
  CuMatrix<Real> T(rows, cols * static_cast<MatrixIndexT>(off.size()));
  cu::Splice(S, offsets, &T);
  Matrix<Real> Ht(T);
  for (MatrixIndexT r = 0; r < rows; r++)
    for (size_t k = 0; k < off.size(); k++) {
      const int32 from = std::min<int32>(rows - 1, std::max<int32>(0, r + off[k]));   // clamped at both ends
      for (MatrixIndexT c = 0; c < cols; c++) KALDI_ASSERT(Ht(r, k * cols + c) == static_cast<Real>(100 * from + c));
    }
}

// ---- CuVector::AddDiagMat2(alpha, M, trans, beta) ------------------------------------------------------------------
template <typename Real>
static void TestAddDiagMat2() {
  const MatrixIndexT rows = 45, cols = 350;
  Matrix<Real> Hm = Pattern<Real>(rows, cols, 0.4, 1.5);
  CuMatrix<Real> M(Hm);
  Vector<Real> hv(rows);
  for (MatrixIndexT r = 0; r < rows; r++) hv(r) = static_cast<Real>(r - 20);
  CuVector<Real> v(hv);
  v.AddDiagMat2(0.5, M, kNoTrans, 2.0);     // v = 2 v + 0.5 diag(M M^T): squared row norms
  Vector<Real> got(rows);
  v.CopyToVec(&got);
  for (MatrixIndexT r = 0; r < rows; r++) {
    double s = 0.0;
    for (MatrixIndexT c = 0; c < cols; c++) s += static_cast<double>(Hm(r, c)) * Hm(r, c);
    EXPECT_NEAR(got(r), 2.0 * (r - 20) + 0.5 * s, Tol<Real>());
  }
  CuVector<Real> w(cols);
  w.AddDiagMat2(1.0, M, kTrans, 0.0);       // diag(M^T M): squared column norms
  Vector<Real> gw(cols);
  w.CopyToVec(&gw);
  for (MatrixIndexT c = 0; c < cols; c += 9) {
    double s = 0.0;
    for (MatrixIndexT r = 0; r < rows; r++) s += static_cast<double>(Hm(r, c)) * Hm(r, c);
    EXPECT_NEAR(gw(c), s, Tol<Real>());
  }
}

// ---- CompObjfAndDeriv(sv_labels, output, &tot_objf, &tot_weight) (float: the discriminative update's arithmetic type) --
static void TestCompObjfAndDeriv() {
  typedef BaseFloat Real;
  const MatrixIndexT rows = 50, cols = 8;
  Matrix<Real> Hout(rows, cols);
  for (MatrixIndexT r = 0; r < rows; r++)
    for (MatrixIndexT c = 0; c < cols; c++) Hout(r, c) = static_cast<Real>((c + 1.0) / 36.0);   // rows sum to one
  CuMatrix<Real> output(Hout), deriv(rows, cols);
  std::vector<MatrixElement<Real> > labels;
  double want_objf = 0.0, want_weight = 0.0;
  for (int32 r = 0; r < rows; r++) {
    const int32 c = (3 * r) % cols;
    const Real w = static_cast<Real>(0.5 + 0.01 * r);
    MatrixElement<Real> e = {r, c, w};
    labels.push_back(e);
    want_objf += w * std::log((c + 1.0) / 36.0);
    want_weight += w;
  }
  Real objf = 0, weight = 0;
  deriv.CompObjfAndDeriv(labels, output, &objf, &weight);
  EXPECT_NEAR(objf, want_objf, 1e-4);
  EXPECT_NEAR(weight, want_weight, 1e-5);
  Matrix<Real> Hd(deriv);
  for (int32 r = 0; r < rows; r++)
    for (int32 c = 0; c < cols; c++) {
      const double want = c == (3 * r) % cols ? (0.5 + 0.01 * r) / ((c + 1.0) / 36.0) : 0.0;   // weight / probability at the label
      EXPECT_NEAR(Hd(r, c), want, 1e-5);
    }
}

// ---- the same primitives on views: the library takes (pointer, rows, cols, stride), a Range() of a larger
// matrix must compute what the owning matrix of the same content computes (cu-matrix.h:447-463) ----------
template <typename Real>
static void UnitTestCuSubMatrixOps() {
  Matrix<Real> Hbig(60, 90);
  Hbig.SetRandn();
  CuMatrix<Real> Dbig(Hbig);
  CuSubMatrix<Real> Dv = Dbig.Range(7, 40, 11, 60);
  SubMatrix<Real> Hv = Hbig.Range(7, 40, 11, 60);
  KALDI_ASSERT(Dv.Stride() == Dbig.Stride() && Dv.NumRows() == 40 && Dv.NumCols() == 60);

  CuMatrix<Real> Dg(40, 12);
  Matrix<Real> Hg(40, 12);
  Dg.GroupPnorm(Dv, 2.0);
  Hg.GroupPnorm(Hv, 2.0);
  Matrix<Real> Hg2(Dg);
  AssertEqual(Hg, Hg2);

  CuMatrix<Real> Dp(40, 40);
  Matrix<Real> Hp(40, 40);
  Dp.AddMatMat(1.0, Dv, kNoTrans, Dv, kTrans, 0.0);
  Hp.AddMatMat(1.0, Hv, kNoTrans, Hv, kTrans, 0.0);
  Matrix<Real> Hp2(Dp);
  AssertEqual(Hp, Hp2);

  // writing through a view leaves the rest of the owner untouched
  Dv.ApplySoftMaxPerRow(Dv);
  Matrix<Real> Hafter(Dbig);
  for (MatrixIndexT r = 0; r < 60; r++)
    for (MatrixIndexT c = 0; c < 90; c++)
      if (r < 7 || r >= 47 || c < 11 || c >= 71) KALDI_ASSERT(Hafter(r, c) == Hbig(r, c));
  for (MatrixIndexT r = 0; r < 40; r++) Hv.Row(r).ApplySoftMax();
  Matrix<Real> Hs(Hafter.Range(7, 40, 11, 60));
  Matrix<Real> Hs_ref(Hv);
  AssertEqual(Hs_ref, Hs, 0.00001);
}

// ---- element access, copies between precisions, Swap (cu-value.h, cu-matrix.cc:128-152, 283-307) ---------
template <typename Real>
static void UnitTestCuMatrixCopyAndValue() {
  Matrix<Real> H(13, 17);
  H.SetRandn();
  CuMatrix<Real> D(H);
  KALDI_ASSERT(Real(D(3, 4)) == H(3, 4));
  D(3, 4) = 7.5;
  D(3, 4) += 0.25;
  KALDI_ASSERT(Real(D(3, 4)) == Real(7.75));
  D(0, 0) = D(3, 4);
  KALDI_ASSERT(Real(D(0, 0)) == Real(7.75));
  const CuMatrix<Real> &Dc = D;
  KALDI_ASSERT(Dc(0, 0) == Real(7.75));

  Matrix<double> Hd(13, 17);
  Hd.SetRandn();
  CuMatrix<Real> Dd(Hd);  // host double -> device Real
  Matrix<double> Hd2(13, 17);
  Dd.CopyToMat(&Hd2);
  for (MatrixIndexT r = 0; r < 13; r++)
    for (MatrixIndexT c = 0; c < 17; c++) KALDI_ASSERT(Hd2(r, c) == static_cast<double>(static_cast<Real>(Hd(r, c))));

  CuMatrix<Real> Dt(H, kTrans);
  KALDI_ASSERT(Dt.NumRows() == 17 && Dt.NumCols() == 13 && Real(Dt(4, 3)) == H(3, 4));

  CuMatrix<Real> A(5, 6), B;
  A.SetRandn();
  Matrix<Real> Ha(A);
  A.Swap(&B);
  KALDI_ASSERT(A.NumRows() == 0 && B.NumRows() == 5 && B.NumCols() == 6);
  Matrix<Real> Hswap(2, 3);
  Hswap(1, 2) = 4.0;
  B.Swap(&Hswap);
  KALDI_ASSERT(B.NumRows() == 2 && Real(B(1, 2)) == Real(4.0) && Hswap.NumRows() == 5);
  AssertEqual(Ha, Hswap, 0.0);

  CuVector<Real> v(9);
  v(2) = 3.0;
  Vector<Real> hv(9);
  v.CopyToVec(&hv);
  KALDI_ASSERT(hv(2) == Real(3.0) && hv(1) == Real(0.0));
  CuSubVector<Real> row(D, 3);
  KALDI_ASSERT(Real(row(4)) == Real(7.75) && row.Dim() == 17);
}

// ---- what stays float-only says so on a <double> object (KALDI_ERR), it does not compute on the host -----------------
static void TestFloatOnlyOperationsRefuseDouble() {
  Matrix<double> H(4, 8);
  H.SetRandn();
  CuMatrix<double> D(H), E(4, 8);
  bool threw = false;
  try {
    E.NormalizePerRow(D);
  } catch (const std::runtime_error &e) {
    threw = std::string(e.what()).find("no double-precision kernel") != std::string::npos;
  }
  KALDI_ASSERT(threw);
}

// ---- LatticeFasterDecoder as lattice-faster-decoder.h:101-140 declares it --------------------------------
static void UnitTestSingleUtteranceDecoder() {
  // 0 -(1:10/0.5)-> 1 -(2:0/0.25)-> 2 (final 0.1); self loop tid 3 on state 1
  std::vector<int64_t> off = {0, 1, 3, 3};
  std::vector<int32> il = {1, 3, 2}, ol = {10, 0, 0}, ns = {1, 1, 2};
  std::vector<float> w = {0.5f, 0.75f, 0.25f}, fin = {INFINITY, INFINITY, 0.1f};
  KhFst *fst = kh_fst_create(3, 0, off.data(), il.data(), ol.data(), w.data(), ns.data(), fin.data());
  KALDI_ASSERT(fst != NULL);

  Matrix<BaseFloat> loglikes(5, 3);
  const BaseFloat rows[5][3] = {{0, -5, -5}, {-5, -5, 0}, {-5, -5, 0}, {-5, -5, 0}, {-5, 0, -5}};
  for (int32 t = 0; t < 5; t++)
    for (int32 j = 0; j < 3; j++) loglikes(t, j) = rows[t][j];
  CuMatrix<BaseFloat> cu_loglikes(loglikes);
  std::vector<int32> tid2pdf_host = {0, 0, 1, 2};  // TransitionIdToPdf, index 0 unused
  CuArray<int32> tid2pdf(tid2pdf_host);

  LatticeFasterDecoderConfig config;
  config.beam = 13.0;
  config.lattice_beam = 6.0;
  LatticeFasterDecoder decoder(*fst, config);
  DecodableMatrixMapped decodable(cu_loglikes, tid2pdf, tid2pdf_host);
  KALDI_ASSERT(decoder.Decode(&decodable));
  KALDI_ASSERT(decoder.ReachedFinal());
  KALDI_ASSERT(decoder.NumFramesDecoded() == 5);
  RawLattice lat;
  KALDI_ASSERT(decoder.GetRawLattice(&lat));
  KALDI_ASSERT(lat.state_frame.size() == 6 && lat.arc_src.size() == 5);
  std::vector<int32> alignment, words;
  BaseFloat graph_cost, acoustic_cost;
  KALDI_ASSERT(decoder.GetBestPath(&alignment, &words, &graph_cost, &acoustic_cost));
  const std::vector<int32> want = {1, 3, 3, 3, 2};
  KALDI_ASSERT(alignment == want && words.size() == 1 && words[0] == 10);
  AssertEqual(graph_cost, 0.5f + 3 * 0.75f + 0.25f + 0.1f, 1e-6);
  KALDI_ASSERT(acoustic_cost == 0.0f);
  AssertEqual(decoder.FinalRelativeCost(), 0.1f, 1e-5);  // best + final(0.1) against best without it

  // a second, longer utterance on the same object: nothing was sized for the first one
  Matrix<BaseFloat> longer(400, 3);
  for (int32 t = 0; t < 400; t++) {
    longer(t, 0) = t == 0 ? 0 : -5;
    longer(t, 2) = (t > 0 && t < 399) ? 0 : -5;
    longer(t, 1) = t == 399 ? 0 : -5;
  }
  CuMatrix<BaseFloat> cu_longer(longer);
  DecodableMatrixMapped decodable2(cu_longer, tid2pdf, tid2pdf_host);
  KALDI_ASSERT(decoder.Decode(&decodable2));
  KALDI_ASSERT(decoder.NumFramesDecoded() == 400 && decoder.ReachedFinal());
  KALDI_ASSERT(decoder.GetBestPath(&alignment, &words, &graph_cost, &acoustic_cost));
  KALDI_ASSERT(alignment.size() == 400 && alignment[0] == 1 && alignment[200] == 3 && alignment[399] == 2);

  // SetOptions + a decodable whose pdf map does not fit the matrix (decodable-matrix.h:41-44's KALDI_ERR)
  config.max_active = 100;
  decoder.SetOptions(config);
  KALDI_ASSERT(decoder.Decode(&decodable));
  std::vector<int32> bad_host = {0, 0, 1, 3};
  CuArray<int32> bad(bad_host);
  bool threw = false;
  try {
    DecodableMatrixMapped d3(cu_loglikes, bad, bad_host);
  } catch (const std::runtime_error &) {
    threw = true;
  }
  KALDI_ASSERT(threw);
  kh_fst_destroy(fst);
}

template <typename Real>
static void CudaMatrixUnitTest() {
  TestAddMatMat<Real>();
  TestSoftmax<Real>();
  TestCopyRows<Real>();
  TestGroupPnorm<Real>();
  TestElementwise<Real>();
  TestRowColumnVectors<Real>();
  TestRangesAndLookup<Real>();
  TestSplice<Real>();
  TestAddDiagMat2<Real>();
  UnitTestCuSubMatrixOps<Real>();
  UnitTestCuMatrixCopyAndValue<Real>();
}

}  // namespace kaldi

int main() {
  using namespace kaldi;
  try {
    // cu-matrix-test.cc:2095-2121 runs the suite with and without a device ("no" / "yes"); this library has no
    // host path, so SelectGpuId("no") must refuse and the suite runs once, on the device
    bool refused = false;
    try {
      CuDevice::Instantiate().SelectGpuId("no");
    } catch (const std::runtime_error &) {
      refused = true;
    }
    KALDI_ASSERT(refused);
    CuDevice::Instantiate().SelectGpuId("yes");
    KALDI_ASSERT(CuDevice::Instantiate().Enabled());
    for (int32 loop = 0; loop < 2; loop++) {
      srand(loop);
      kaldi::CudaMatrixUnitTest<float>();
      KALDI_ASSERT(CuDevice::Instantiate().DoublePrecisionSupported());
      kaldi::CudaMatrixUnitTest<double>();   // (the reference runs both instantiations, cu-matrix-test.cc:2100-2114)
      TestCompObjfAndDeriv();
      TestFloatOnlyOperationsRefuseDouble();
    }
    UnitTestSingleUtteranceDecoder();
    CuDevice::Instantiate().PrintProfile();
  } catch (const std::exception &e) {
    fprintf(stderr, "FAILED: %s\n", e.what());
    return 1;
  }
  printf("all tests passed\n");
  return 0;
}
