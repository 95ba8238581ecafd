// feature_oracle.cc — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the feature front-end next to the hot path:
//   NumFrames / ExtractWindow / Preemphasize / FeatureWindowFunction / ComputePowerSpectrum
//                              feat/feature-functions.cc:29-48,61-70,74-92,98-167,186-207
//   MelBanks (vtln_warp = 1)   feat/mel-computations.cc:33-152, Compute :219-246
//   Mfcc::ComputeInternal      feat/feature-mfcc.cc:119-184 (use_energy = false, no htk_compat)
//   ComputeDctMatrix           matrix/matrix-functions.cc:592-608, ComputeLifterCoeffs mel-computations.cc:248-254
//   DeltaFeatures              feat/feature-functions.cc:210-267, ComputeDeltas :361-372
//   AccCmvnStats / ApplyCmvn   transform/cmvn.cc:30-113
// The FFT (the reference's SplitRadixRealFft, matrix/srfft.cc) is restated as its
// definition, a DFT accumulated in double; dither = 0 (the only random step).
// PINNED against the reference's own code (oracle/_ref, feat/ + transform/cmvn.cc compiled
// where they lie) by tests/test_oracle_vs_ref.py and the golden vectors.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

namespace {
const double kPi = 3.1415926535897932384626433832795, k2Pi = 6.283185307179586476925286766559;

inline float MelScale(float freq) { return 1127.0f * logf(1.0f + freq / 700.0f); }

struct MelBin { int first; std::vector<float> w; };

int RoundUpPow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }
}  // namespace

extern "C" {

// NumFrames feature-functions.cc:29-48
int ko_num_frames(int nsamp, int frame_shift, int frame_length, int snip_edges) {
  if (snip_edges) return nsamp < frame_length ? 0 : 1 + (nsamp - frame_length) / frame_shift;
  return static_cast<int>(nsamp * 1.0f / frame_shift + 0.5f);
}

// The tables Mfcc's constructor builds: window (frame_length), mel bins (first index,
// offsets, weights), DCT rows (num_ceps x num_bins), lifter (num_ceps).  mel_weights must
// hold num_bins * (padded / 2) floats at most; returns the number of weights written.
int ko_mfcc_tables(float samp_freq, float frame_length_ms, float frame_shift_ms, const char *window_type,
                   int num_bins, float low_freq, float high_freq_opt, int num_ceps, float cepstral_lifter,
                   int *frame_shift_out, int *frame_length_out, int *padded_out, float *window, int32_t *mel_first,
                   int32_t *mel_off, float *mel_weights, float *dct, float *lifter) {
  const int frame_shift = static_cast<int>(samp_freq * 0.001f * frame_shift_ms),     // WindowShift() .h:119-121
            frame_length = static_cast<int>(samp_freq * 0.001f * frame_length_ms);   // WindowSize()
  const int padded = RoundUpPow2(frame_length);
  *frame_shift_out = frame_shift; *frame_length_out = frame_length; *padded_out = padded;
  const std::string wt(window_type);
  for (int i = 0; i < frame_length; i++) {  // FeatureWindowFunction :74-92
    const float i_fl = static_cast<float>(i);
    if (wt == "hanning") window[i] = 0.5 - 0.5 * cos(k2Pi * i_fl / (frame_length - 1));
    else if (wt == "hamming") window[i] = 0.54 - 0.46 * cos(k2Pi * i_fl / (frame_length - 1));
    else if (wt == "povey") window[i] = pow(0.5 - 0.5 * cos(k2Pi * i_fl / (frame_length - 1)), 0.85);
    else window[i] = 1.0;
  }
  // MelBanks :33-152 with vtln_warp_factor == 1.0
  const int window_length = static_cast<int>(samp_freq * 0.001 * frame_length_ms);
  const int window_length_padded = RoundUpPow2(window_length);
  const int num_fft_bins = window_length_padded / 2;
  const float nyquist = 0.5 * samp_freq;
  const float high_freq = high_freq_opt > 0.0 ? high_freq_opt : nyquist + high_freq_opt;
  const float fft_bin_width = samp_freq / window_length_padded;
  const float mel_low_freq = MelScale(low_freq), mel_high_freq = MelScale(high_freq);
  const float mel_freq_delta = (mel_high_freq - mel_low_freq) / (num_bins + 1);
  int nw = 0;
  mel_off[0] = 0;
  for (int bin = 0; bin < num_bins; bin++) {
    const float left_mel = mel_low_freq + bin * mel_freq_delta, center_mel = mel_low_freq + (bin + 1) * mel_freq_delta,
                right_mel = mel_low_freq + (bin + 2) * mel_freq_delta;
    std::vector<float> this_bin(num_fft_bins, 0.0f);
    int first_index = -1, last_index = -1;
    for (int i = 0; i < num_fft_bins; i++) {
      const float freq = fft_bin_width * i;
      const float mel = MelScale(freq);
      if (mel > left_mel && mel < right_mel) {
        float weight;
        if (mel <= center_mel) weight = (mel - left_mel) / (center_mel - left_mel);
        else weight = (right_mel - mel) / (right_mel - center_mel);
        this_bin[i] = weight;
        if (first_index == -1) first_index = i;
        last_index = i;
      }
    }
    if (first_index == -1) return -1;  // "You may have set --num-mel-bins too large."
    mel_first[bin] = first_index;
    for (int i = first_index; i <= last_index; i++) mel_weights[nw++] = this_bin[i];
    mel_off[bin + 1] = nw;
  }
  // ComputeDctMatrix matrix-functions.cc:592-608 on a num_bins x num_bins matrix, first num_ceps rows
  {
    const int N = num_bins;
    float normalizer = std::sqrt(1.0 / static_cast<float>(N));
    for (int j = 0; j < N; j++) dct[j] = normalizer;
    normalizer = std::sqrt(2.0 / static_cast<float>(N));
    for (int k = 1; k < num_ceps; k++)
      for (int n = 0; n < N; n++) dct[k * N + n] = normalizer * std::cos(static_cast<double>(kPi) / N * (n + 0.5) * k);
  }
  for (int i = 0; i < num_ceps; i++)  // ComputeLifterCoeffs mel-computations.cc:248-254
    lifter[i] = cepstral_lifter != 0.0f ? 1.0 + 0.5 * cepstral_lifter * sin(kPi * i / cepstral_lifter) : 1.0f;
  return nw;
}

// Mfcc::ComputeInternal feature-mfcc.cc:119-184, dither = 0 (the only random step).
// snip_edges = 0: frames centred on frame_shift * (r + 0.5), the signal extended by reflection
// (ExtractWindow :107-135); use_energy: C0 replaced by the log energy, before pre-emphasis and
// windowing (raw_energy) or of the windowed frame, floored at log(energy_floor) (:138-141,
// :167-171); htk_compat: the energy / C0 * sqrt(2) moved to the last column (:173-182).
int ko_mfcc_compute_opts(const float *wave, int n_samples, float samp_freq, float frame_length_ms, float frame_shift_ms,
                         float preemph_coeff, int remove_dc_offset, const char *window_type, int snip_edges, int use_energy,
                         int raw_energy, float energy_floor, int htk_compat, int num_bins, float low_freq, float high_freq,
                         int num_ceps, float cepstral_lifter, float *out, int out_stride, int max_rows) {
  int frame_shift, frame_length, padded;
  std::vector<float> window(RoundUpPow2(static_cast<int>(samp_freq * 0.001f * frame_length_ms)) + 8),
      weights(static_cast<size_t>(num_bins) * window.size()), dct(static_cast<size_t>(num_ceps) * num_bins), lifter(num_ceps);
  std::vector<int32_t> first(num_bins), off(num_bins + 1);
  if (ko_mfcc_tables(samp_freq, frame_length_ms, frame_shift_ms, window_type, num_bins, low_freq, high_freq, num_ceps,
                     cepstral_lifter, &frame_shift, &frame_length, &padded, window.data(), first.data(), off.data(),
                     weights.data(), dct.data(), lifter.data()) < 0)
    return -2;
  const int rows = ko_num_frames(n_samples, frame_shift, frame_length, snip_edges);
  if (rows > max_rows) return -1;
  std::vector<float> win(padded), power(padded / 2 + 1), mel(num_bins);
  const float fmin = std::numeric_limits<float>::min();
  for (int r = 0; r < rows; r++) {
    // ExtractWindow :98-167
    if (snip_edges) {
      const float *w0 = wave + static_cast<size_t>(frame_shift) * r;
      for (int i = 0; i < frame_length; i++) win[i] = w0[i];
    } else {
      const int mid = static_cast<int>(frame_shift * (r + 0.5)), begin = mid - frame_length / 2;
      for (int i = 0; i < frame_length; i++) {
        const int f = begin + i;
        int src = f;
        if (f < 0) src = (-f) % n_samples;
        else if (f >= n_samples) src = n_samples - 1 - (f - n_samples) % n_samples;
        win[i] = wave[src];
      }
    }
    if (remove_dc_offset) {  // window_part.Add(-window_part.Sum() / frame_length)
      double dsum = 0.0;     // VectorBase::Sum kaldi-vector.cc: double accumulation, float result
      for (int i = 0; i < frame_length; i++) dsum += win[i];
      const float sum = static_cast<float>(dsum);
      const float c = -sum / frame_length;
      for (int i = 0; i < frame_length; i++) win[i] += c;
    }
    float log_energy = 0.0f;
    if (use_energy && raw_energy) {  // :151-155 (VecVec: float dot product)
      float e = 0.0f;
      for (int i = 0; i < frame_length; i++) e += win[i] * win[i];
      log_energy = logf(e > fmin ? e : fmin);
    }
    if (preemph_coeff != 0.0f) {  // Preemphasize :61-67
      for (int i = frame_length - 1; i > 0; i--) win[i] -= preemph_coeff * win[i - 1];
      win[0] -= preemph_coeff * win[0];
    }
    for (int i = 0; i < frame_length; i++) win[i] *= window[i];
    for (int i = frame_length; i < padded; i++) win[i] = 0.0f;
    if (use_energy && !raw_energy) {  // feature-mfcc.cc:140-142
      float e = 0.0f;
      for (int i = 0; i < padded; i++) e += win[i] * win[i];
      log_energy = logf(e > fmin ? e : fmin);
    }
    // srfft_->Compute + ComputePowerSpectrum :186-207: |X_k|^2, k = 0 .. N/2
    for (int k = 0; k <= padded / 2; k++) {
      double re = 0.0, im = 0.0;
      for (int n = 0; n < padded; n++) {
        const double a = k2Pi * ((static_cast<long long>(k) * n) % padded) / padded;
        re += win[n] * cos(a);
        im -= win[n] * sin(a);
      }
      const float fre = static_cast<float>(re), fim = static_cast<float>(im);
      power[k] = fre * fre + fim * fim;
    }
    // MelBanks::Compute :219-246 (VecVec), floor, log
    for (int b = 0; b < num_bins; b++) {
      float e = 0.0f;
      for (int i = off[b]; i < off[b + 1]; i++) e += weights[i] * power[first[b] + (i - off[b])];
      if (e < fmin) e = fmin;
      mel[b] = logf(e);
    }
    // this_mfcc = dct_matrix_ * mel_energies; MulElements(lifter)
    float *row = out + static_cast<size_t>(r) * out_stride;
    for (int c = 0; c < num_ceps; c++) {
      float s = 0.0f;
      for (int b = 0; b < num_bins; b++) s += dct[c * num_bins + b] * mel[b];
      if (cepstral_lifter != 0.0f) s *= lifter[c];
      row[c] = s;
    }
    if (use_energy) {  // :167-171
      if (energy_floor > 0.0f && log_energy < logf(energy_floor)) log_energy = logf(energy_floor);
      row[0] = log_energy;
    }
    if (htk_compat) {  // :173-182
      float energy = row[0];
      for (int i = 0; i < num_ceps - 1; i++) row[i] = row[i + 1];
      if (!use_energy) energy *= static_cast<float>(M_SQRT2);
      row[num_ceps - 1] = energy;
    }
  }
  return rows;
}

int ko_mfcc_compute(const float *wave, int n_samples, float samp_freq, float frame_length_ms, float frame_shift_ms,
                    float preemph_coeff, int remove_dc_offset, const char *window_type, int num_bins, float low_freq,
                    float high_freq, int num_ceps, float cepstral_lifter, float *out, int out_stride, int max_rows) {
  return ko_mfcc_compute_opts(wave, n_samples, samp_freq, frame_length_ms, frame_shift_ms, preemph_coeff, remove_dc_offset,
                              window_type, 1, 0, 1, 0.0f, 0, num_bins, low_freq, high_freq, num_ceps, cepstral_lifter, out,
                              out_stride, max_rows);
}

// DeltaFeatures scales :210-242; returns the length of scales[order] (all orders are
// written into `scales` back to back, lengths into `lens`).
void ko_delta_scales(int order, int window, float *scales, int32_t *lens) {
  std::vector<std::vector<float> > sc(order + 1);
  sc[0].assign(1, 1.0f);
  for (int i = 1; i <= order; i++) {
    const std::vector<float> &prev = sc[i - 1];
    std::vector<float> &cur = sc[i];
    const int prev_offset = (static_cast<int>(prev.size()) - 1) / 2, cur_offset = prev_offset + window;
    cur.assign(prev.size() + 2 * window, 0.0f);
    float normalizer = 0.0f;
    for (int j = -window; j <= window; j++) {
      normalizer += j * j;
      for (int k = -prev_offset; k <= prev_offset; k++) cur[j + k + cur_offset] += static_cast<float>(j) * prev[k + prev_offset];
    }
    const float inv = static_cast<float>(1.0 / normalizer);  // Scale(BaseFloat alpha): the double quotient becomes a float argument
    for (size_t k = 0; k < cur.size(); k++) cur[k] = cur[k] * inv;
  }
  int pos = 0;
  for (int i = 0; i <= order; i++) {
    lens[i] = static_cast<int32_t>(sc[i].size());
    for (float v : sc[i]) scales[pos++] = v;
  }
}

// ComputeDeltas :361-372 = DeltaFeatures::Process :244-267 for every frame
void ko_compute_deltas(const float *in, int rows, int cols, int in_stride, int order, int window, float *out, int out_stride) {
  std::vector<float> scales((order + 1) * (2 * window * order + 1));
  std::vector<int32_t> lens(order + 1);
  ko_delta_scales(order, window, scales.data(), lens.data());
  for (int t = 0; t < rows; t++) {
    int pos = 0;
    for (int i = 0; i <= order; i++) {
      const int max_offset = (lens[i] - 1) / 2;
      float *o = out + static_cast<size_t>(t) * out_stride + i * cols;
      for (int d = 0; d < cols; d++) o[d] = 0.0f;
      for (int j = -max_offset; j <= max_offset; j++) {
        int f = t + j;
        if (f < 0) f = 0; else if (f >= rows) f = rows - 1;
        const float scale = scales[pos + j + max_offset];
        if (scale != 0.0f)
          for (int d = 0; d < cols; d++) o[d] += scale * in[static_cast<size_t>(f) * in_stride + d];  // AddVec
      }
      pos += lens[i];
    }
  }
}

// AccCmvnStats transform/cmvn.cc:30-62 (weights == NULL); stats [2 x (cols + 1)] doubles
void ko_acc_cmvn_stats(const float *feats, int rows, int cols, int stride, double *stats) {
  for (int r = 0; r < rows; r++) {
    stats[cols] += 1.0f;
    for (int d = 0; d < cols; d++) {
      const float x = feats[static_cast<size_t>(r) * stride + d];
      stats[d] += x * 1.0f;                    // *mean_ptr += *feats_ptr * weight (float product)
      stats[cols + 1 + d] += x * x * 1.0f;     // *var_ptr += *feats_ptr * *feats_ptr * weight
    }
  }
}

// ApplyCmvn transform/cmvn.cc:64-113; returns -1 if count < 1
int ko_apply_cmvn(const double *stats, int var_norm, float *feats, int rows, int cols, int stride) {
  const double count = stats[cols];
  if (count < 1.0) return -1;
  std::vector<float> offset(cols), scale(cols);
  for (int d = 0; d < cols; d++) {
    double mean = stats[d] / count, off, sc;
    if (!var_norm) { sc = 1.0; off = -mean; }
    else {
      double var = (stats[cols + 1 + d] / count) - mean * mean;
      if (var < 1.0e-20) var = 1.0e-20;
      sc = 1.0 / sqrt(var);
      off = -(mean * sc);
    }
    offset[d] = static_cast<float>(off);
    scale[d] = static_cast<float>(sc);
  }
  for (int r = 0; r < rows; r++)
    for (int d = 0; d < cols; d++) {
      float &x = feats[static_cast<size_t>(r) * stride + d];
      if (var_norm) x = x * scale[d];   // MulColsVec
      x = x + 1.0f * offset[d];         // AddVecToRows(1.0, offset)
    }
  return 0;
}

}  // extern "C"
