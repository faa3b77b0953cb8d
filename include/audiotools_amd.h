/* audiotools_amd.h -- C ABI of libaudiotools_amd.so (AMD Instinct MI355X / gfx950).
 *
 * The reference (descriptinc/audiotools v0.7.4) is pure Python and has no FFI seam;
 * its hot path reaches native code only through torch / julius / torchaudio calls.
 * Each entry point below replaces ONE such call site (file:line relative to
 * /root/reference) and is what a binding in the reference's host language (Python
 * ctypes) would bind.
 *
 * Conventions
 *  - plain pointers and sizes; device pointers unless a name ends in _host;
 *  - the caller owns every buffer (torch allocates them in our host layer);
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *    enqueued asynchronously on it, nothing synchronises the device;
 *  - return 0 on success, AT_ERR_INVALID (-1) for a bad argument,
 *    AT_ERR_UNSUPPORTED (-2) when this entry point has no kernel for a valid request,
 *    -1000-hipError_t for a HIP runtime failure.  No exceptions cross the ABI.
 *  - row = one (batch item, channel) pair; audio is (rows, T) float32 contiguous.
 */
#ifndef AUDIOTOOLS_AMD_H
#define AUDIOTOOLS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AT_OK 0
#define AT_ERR_INVALID (-1)
#define AT_ERR_UNSUPPORTED (-2)

/* padding_type of AudioSignal.stft (torch.nn.functional.pad modes, audio_signal.py:1192) */
#define AT_PAD_REFLECT 0
#define AT_PAD_CONSTANT 1
#define AT_PAD_REPLICATE 2
#define AT_PAD_CIRCULAR 3

/* ---- STFT (+ mel) -------------------------------------------------------------------
 * Replaces  audiotools/core/audio_signal.py:1192-1202  F.pad + torch.stft(center=True,
 *           return_complex=True) and  :1355-1368  torch.abs + (mag^T @ mel_basis^T).
 *
 * at_stft_twiddles_host: fills out_host[2*n_fft] with (cos, -sin)(2*pi*k/n_fft), k < n_fft
 *   (float64 math, rounded to float32).  Upload once per n_fft and pass as `twiddles`.
 * at_stft_fused_supported:  1 if n_fft is a power of two in [32, 2048] (fused wave-FFT kernels: mel stage,
 *                           register reuse, adjoint kernels).
 * at_stft_native_supported: 1 if SOME native kernel covers n_fft: the fused ones, or the generic mixed-radix
 *                           transform (even n_fft <= 16384 with n_fft/2 = 2^a 3^b 5^c 7^d: 4096, 8192, 400, 1200,
 *                           1920 ...).  Those sizes take the mel tables in BANDED form (at_mel_bands_host) and
 *                           fuse the mel stage for n_fft <= 8192; mel_out must be NULL beyond that.
 * at_stft_mel_f32:
 *   x          (rows, T) f32
 *   window     (n_fft) f32            AudioSignal.get_window (audio_signal.py:1009-1039)
 *   pad, right_pad, pad_mode          match_stride outer padding (audio_signal.py:1089-1121)
 *   frame_lo, n_frames_out            frames [frame_lo, frame_lo+n_frames_out) of the
 *                                     1 + (T+2*pad+right_pad)/hop frames are produced
 *                                     (match_stride drops 2 at each end, :1206-1209)
 *   stft_out   (rows, n_frames_out, n_fft/2+1) complex64 as interleaved f32 (required)
 *              -- bin-contiguous, the physical layout of torch.stft's result
 *   mel tables banded filterbank in "unit" form built by at_mel_units_host: a unit is one
 *              (row of 16 bins, band) pair: mel_unit_info[2*n_units] i32, mel_unit_w[16*n_units] f32,
 *              n_units in {128, 256, 384} (padded by the helper)
 *   mel_out    (rows, n_frames_out, n_mels) f32, or NULL
 *
 *              For the generic sizes (at_stft_fused_supported(n_fft) == 0): the BANDED form built by
 *              at_mel_bands_host -- every filterbank row cut to its non-zero span and the span into chunks of 16
 *              bins: mel_unit_info = n_units first bins (one per chunk) followed by {first chunk, count} per band
 *              (n_units + 2 * n_mels i32), mel_unit_w = (n_units, 16) zero-padded chunk weights.
 *
 * at_mel_bands_host: HOST helper for that banded form (audio_signal.py:1355-1368: abs + matmul with a basis
 *   whose rows are >= 97 % zeros).  Call with info == NULL for the chunk count n, then with info[n + 2 * n_mels]
 *   and w[16 * n].
 * at_mel_units_host: HOST helper that compresses a dense (n_mels, n_bins) float32 filterbank
 *   (librosa.filters.mel layout, audio_signal.py:1323-1331) into unit tables.  Call with
 *   unit_info == NULL to get the unit count, then again with buffers.  Returns the count (>0)
 *   or a negative error.
 */
int at_stft_twiddles_host(int n_fft, float* out_host);
int at_stft_native_supported(int n_fft);
int at_stft_fused_supported(int n_fft);
int at_mel_units_host(const float* basis_host, int n_mels, int n_bins, int* unit_info_host, float* unit_w_host);
int at_mel_bands_host(const float* basis_host, int n_mels, int n_bins, int* info_host, float* w_host);
int at_stft_mel_f32(const float* x, int64_t rows, int64_t T, const float* window, const float* twiddles,
                    int n_fft, int hop, int pad, int right_pad, int pad_mode, int frame_lo, int64_t n_frames_out,
                    float* stft_out, const int* mel_unit_info, const float* mel_unit_w, int n_units, int n_mels,
                    float* mel_out, void* stream);

/* (The measurement twin of at_stft_mel_f32 -- same arguments, the kernel's loads and stores without its arithmetic, what
 * bench.py prints as roofline.floor_ms -- is an entry point of the DEVELOPMENT build only since round 6:
 * lib/libaudiotools_amd_dev.so, `python -m audiotools_amd._native --dev`; audiotools_amd/_native.py: DEV_SIGNATURES.) */

/* ---- inverse STFT --------------------------------------------------------------------------
 * Replaces  audiotools/core/audio_signal.py:1278-1290  (F.pad of the two edge frames when
 *           match_stride) + torch.istft(X, n_fft, hop, window, length, center=True).
 *           Power-of-two n_fft in [32, 2048]; any hop.
 *   X        (rows, n_x, n_fft/2+1) complex64 interleaved, bin-contiguous (the layout
 *            at_stft_mel_f32 writes / torch.stft returns)
 *   lead, n_frames   the transform runs over n_frames >= lead + n_x VIRTUAL frames: `lead` all-zero
 *            frames, the n_x frames of X, then all-zero frames (match_stride: lead = 2,
 *            n_frames = n_x + 4; otherwise lead = 0, n_frames = n_x)
 *   out      (rows, length) f32; samples with no frame overlap are 0 (torch pads to `length`)
 *   workspace at_istft_workspace_bytes(rows, n_frames, n_fft, hop) bytes
 * hop = n_fft/{2,4,8,16}: ONE fused kernel (FFT, window, overlap-add in registers, envelope
 * division) -- X is read once, out written once; the workspace only holds the reciprocal envelope.
 * Other hops go through a (rows, n_frames, n_fft) frame buffer in the workspace (lead must be 0).
 */
int64_t at_istft_workspace_bytes(int64_t rows, int64_t n_frames, int n_fft, int hop);
int at_istft_f32(const float* X, int64_t rows, int64_t n_x, const float* window, const float* twiddles, int n_fft,
                 int hop, int lead, int64_t n_frames, int64_t length, float* out, void* workspace,
                 int64_t workspace_bytes, void* stream);
/* at_istft_f32 of edit(X): the STFT-domain edits of audiotools/core/dsp.py:217-352 (mask_frequencies,
 * mask_timesteps, shift_phase, mask_low_magnitudes; per-ITEM parameters, item = row / C) applied to the spectrum as the
 * inverse transform reads it -- SpectralTransform.transform (data/transforms.py:274-286) is stft -> edit -> istft, and
 * the edit's own read + write of stft_data disappears.  X is not modified.
 *   kind 1: bins  lo[b] <= k < hi[b] := (fill_re, fill_im);   kind 2: frames lo[b] <= n < hi[b] := fill;
 *   kind 3: X * e^{i shift[b]};   kind 4: 10 log10(max(|X|^2, 1e-10)) [floored at max - top_db] < cut_db[b]
 *           -> val e^{i angle X}  (maxpow: 1 float, the batch maximum of |X|^2 from at_spec_maxpow_f32).
 * hop must be n_fft / 4, 64 <= n_fft <= 2048; AT_ERR_UNSUPPORTED otherwise (edit with at_spec_*, then at_istft_f32). */
int at_istft_edit_f32(const float* X, int64_t rows, int64_t n_x, const float* window, const float* twiddles, int n_fft,
                      int hop, int lead, int64_t n_frames, int64_t length, float* out, void* workspace,
                      int64_t workspace_bytes, int kind, int64_t C, const int* lo, const int* hi, const float* shift,
                      const double* cut_db, const float* maxpow, float fill_re, float fill_im, float top_db,
                      int use_top_db, float val, void* stream);

/* ---- adjoint of the forward STFT (backward pass of stft()) --------------------------------
 * What torch.autograd computes for  audiotools/core/audio_signal.py:1195  torch.stft(center=True)
 * (used by the reference's training losses, audiotools/metrics/spectral.py).
 *   G    (rows, n_frames, n_fft/2+1) complex64 = dL/dX as autograd hands it, bin-contiguous
 *   out  (rows, out_len) f32, out_len >= Lp = (n_frames-1)*hop + n_fft: gradient w.r.t. the
 *        CENTRE-PADDED signal (positions >= Lp are not written)
 *        out[n] = sum_f window[n - f hop] * sum_k Re(G[f,k] e^{+2 pi i k (n - f hop) / n_fft});
 *        the caller folds the two reflected margins of n_fft/2 samples back onto the signal.
 * Same fused kernel as at_istft_f32 (no 1/N, DC/Nyquist weighting of the adjoint, no envelope).
 * hop must be n_fft / {2,4,8,16} (AT_ERR_UNSUPPORTED otherwise).
 */
int at_stft_adjoint_f32(const float* G, int64_t rows, int64_t n_frames, const float* window, const float* twiddles,
                        int n_fft, int hop, float* out, int64_t out_len, void* stream);

/* ---- backward of the fused STFT + mel path ---------------------------------------------------
 * mel = basis . |X| (audio_signal.py:1355-1368) under autograd (metrics/spectral.py
 * MelSpectrogramLoss): given ONLY dL/dmel, the spectrum gradient
 *     G[f,k] = (sum_m basis[m,k] gmel[f,m]) * X[f,k] / |X[f,k]|        (0 where X == 0)
 * is formed inside the adjoint kernel from the saved spectrum and never written to memory.
 *   X (rows, n_frames, n_fft/2+1) complex64 saved by the forward pass; gmel (rows, n_frames, n_mels)
 *   bin_bands (n_fft/2+1) i32 = lo | hi << 16 and bin_w (n_fft/2+1, 2) f32: the at most two bands of
 *   a triangular filterbank that cover each bin and their weights (0 where absent)
 *   out as at_stft_adjoint_f32.   hop == n_fft/4, 64 <= n_fft <= 2048, n_mels <= n_fft/4.
 */
int at_stft_mel_adjoint_f32(const float* X, const float* gmel, const int* bin_bands, const float* bin_w, int n_mels,
                            int64_t rows, int64_t n_frames, const float* window, const float* twiddles, int n_fft,
                            int hop, float* out, int64_t out_len, void* stream);

/* ---- BS.1770 integrated loudness ---------------------------------------------------
 * Replaces  audiotools/core/loudness.py:102-126  (2x torchaudio.functional.lfilter, the
 *           CPU/IIR branch), :164-174 (julius.core.unfold), :176-247 (gated integration).
 *   x         (B, C, T) f32, C <= 5
 *   sos_host  nstage x 6 float64 (b0 b1 b2 a0 a1 a2), HOST memory, applied in order;
 *             rounded to float32 inside, as loudness.py:118-119 does
 *   gains_host nstage passband gains (float64, HOST)
 *   K, S      block and stride in samples: int(T_g*rate), int(T_g*rate*0.25)
 *   inv_norm  1/(T_g*rate)
 *   floor_db  clamp of the result (LoudnessMixin.MIN_LOUDNESS = -70), NaN = no clamp
 *   warm      samples a row segment is started early so the filter transient is gone
 *   out       (B) f32 LUFS
 *   workspace at_lufs_workspace_bytes(...) bytes of device scratch
 */
int64_t at_lufs_workspace_bytes(int64_t B, int64_t C, int64_t T, int K, int S);
int at_lufs_f32(const float* x, int64_t B, int64_t C, int64_t T, const double* sos_host, const double* gains_host,
                int nstage, int K, int S, double inv_norm, float floor_db, int warm, float* out, void* workspace,
                int64_t workspace_bytes, void* stream);

/* ---- per-item FIR (low_pass / high_pass / equalizer) ------------------------------------
 * Replaces  audiotools/core/dsp.py:177-179, 209-211  (Python loop over the batch building a
 *           julius.LowPassFilter / HighPassFilter per item) and  audiotools/core/effects.py:399-403,
 *           429-432  (julius.SplitBands + weighted band sum, collapsed to one composite FIR per item).
 *   x      (B, C, T) f32
 *   taps   (taps_rows, L_padded) f32, taps_rows == 1 (shared) or B (per item); every row is an
 *          odd-length FIR centred at index `half`, zero-padded to L_padded (multiple of 8)
 *   out    (B, C, T): cross-correlation with replicate padding; highpass != 0 gives x - FIR(x)
 */
int at_fir_per_item_f32(const float* x, int64_t B, int64_t C, int64_t T, const float* taps, int taps_rows,
                        int L_padded, int half, int highpass, float* out, void* stream);

/* Same filter evaluated by overlap-save block FFTs (2048-sample blocks, wave-level FFT shared with
 * the STFT kernel): the cost per output does not grow with the tap count (the direct form above
 * spends 2*taps flop per sample).  Identical arguments and semantics; additionally
 *   twiddles2048  (2048, 2) f32 device copy of at_stft_twiddles_host(2048, .)
 * L_padded need not be a multiple of 8 here.  Filters longer than 1536 taps run as partitions of
 * 1024 taps (one launch each, accumulating into out).  x and out must not alias.
 */
int at_fir_fft_f32(const float* x, int64_t B, int64_t C, int64_t T, const float* taps, int taps_rows, int L_padded,
                   int half, int highpass, const float* twiddles2048, float* out, void* stream);

/* ---- per-item filter design on the device ------------------------------------------------
 * Replaces the per-item julius.LowPassFilter construction of  audiotools/core/dsp.py:177-179, 209-211  and the
 * julius.SplitBands + weighted band sum of  audiotools/core/effects.py:399-403, 429-432  (host loops over the batch in the
 * reference; ~25 / ~8 whole-table torch launches per call in this package before round 3): ONE launch writes the
 * zero-padded (B, L_padded) tap table at_fir_fft_f32 / at_fir_per_item_f32 read.
 *   at_sinc_taps_f32: cutoffs (B) f32 normalised cutoffs in [0, 0.5] (0 = all-zero filter; the range checks and
 *     H = max_b int(zeros / c_b / 2) are the caller's, from its host copy); row b = Hann-windowed sinc of half size
 *     int(zeros / c_b / 2), unit DC gain, centred at column H; L_padded >= 2 H + 1.
 *   at_eq_taps_f32: weights (B, n_bands) linear gains, bank (n_bands - 1, L) low-pass bank of the band split (centre
 *     `half`); taps[b] = sum_k (w[b,k] - w[b,k+1]) bank[k] + w[b,last] delta(half); B <= 65535.
 */
int at_sinc_taps_f32(const float* cutoffs, int64_t B, float zeros, int H, int L_padded, float* taps, void* stream);
int at_eq_taps_f32(const float* weights, const float* bank, int64_t B, int n_bands, int L, int half, int L_padded, float* taps,
                   void* stream);

/* ---- in-place edits of stft_data ---------------------------------------------------------
 * Replaces the polar round trips (abs, angle, masked_fill, exp, multiply: 8-10 whole-tensor passes)
 * of  audiotools/core/dsp.py:217-306 (mask_frequencies / mask_timesteps),  :308-334
 * (mask_low_magnitudes, with log_magnitude of audio_signal.py:1457-1487 incl. its batch-global
 * top_db floor)  and  :336-352 (shift_phase with one value per item).
 *   X  (B, C, N, F) complex64 interleaved, bin-contiguous: the result.  src: the input spectrum of
 *      the same layout (out of place: one read + one write instead of clone + edit), or NULL / == X
 *      for in place (only the masked region is written).  Unmasked elements are copied unchanged
 *      (the reference rewrites them as |X| e^{i angle X} = X up to rounding)
 *   at_spec_mask_f32: axis 0 masks bins f with lo[b] <= grid[f] < hi[b]; axis 1 masks frames n with
 *      lo[b] <= grid[n] < hi[b]; lo/hi (B) float64, grid float32 (torch.linspace of the reference);
 *      masked elements := (fill_re, fill_im) = val e^{i val}
 *   at_spec_maxpow_f32: *out (device float) = max |X|^2;  at_spec_mask_lowmag_f32 consumes it
 */
int at_spec_mask_f32(const float* src, float* X, int64_t B, int64_t C, int64_t N, int64_t F, int axis, const double* lo,
                     const double* hi, const float* grid, float fill_re, float fill_im, void* stream);
int at_spec_phase_shift_f32(const float* src, float* X, int64_t B, int64_t C, int64_t N, int64_t F, const float* shift,
                            void* stream);
/* per-ELEMENT polar edits (operands a, b in the reference's logical (B, C, F, N) layout):
 *   mode 0: X = src * e^{i b}                  audiotools/core/dsp.py:354-370 corrupt_phase, data/transforms.py:1250-1278
 *   mode 1: X = (src == 0) ? a e^{i b} : src   data/transforms.py:1456-1536 TimeNoise / FrequencyNoise refill */
int at_spec_polar_elem_f32(const float* src, float* X, int64_t B, int64_t C, int64_t N, int64_t F, const float* a,
                           const float* b, int mode, void* stream);
int at_spec_maxpow_f32(const float* X, int64_t n, float* out, void* stream);
/* Spectral gate core (audiotools/ml/layers/spectral_gate.py:96-121):
 *   gate = 20 log10(max(|X|, 1e-4)) < thr_db;  Y = X * (1 - amount[b] * conv2d(gate, outer(tf, tt), zero padding))
 * in ONE pass (gate bits of 16 + halo frames in LDS, separable tent smoothing in registers) instead of ~8 torch passes.
 * X, Y (B, C, N, F) complex64 bin-contiguous; thr_db (B*C, F) if thr_per_item else (C, F); amount (B); tf (kf), tt (kt)
 * odd lengths <= 17 whose outer product is the normalised smoothing filter. */
int at_spec_gate_f32(const float* X, float* Y, int64_t B, int64_t C, int64_t N, int64_t F, const float* thr_db,
                     int thr_per_item, const float* amount, const float* tf, int kf, const float* tt, int kt, void* stream);
int at_spec_mask_lowmag_f32(const float* src, float* X, int64_t B, int64_t C, int64_t N, int64_t F, const double* cutoff_db,
                            const float* maxpow, float top_db, int use_top_db, float val, void* stream);

/* ---- phase vocoder (time_stretch / pitch_shift) --------------------------------------------------
 * Replaces audiotools/core/effects.py:247-309: the reference pipes the batch through CPU libsox
 * ("tempo" / "pitch" + "rate", torchaudio.sox_effects.apply_effects_tensor).  Device-side
 * behavioural equivalent (sox's WSOLA is not reproducible sample for sample; parity is by
 * properties -- length, pitch ratio, batched == single):  stft -> at_phase_vocoder_f32 -> istft,
 * plus at_resample_f32 for pitch_shift.
 *   X (rows, n_in, F) complex64 (physical layout of stft_data), rate = p / q as an exact rational
 *   Y (rows, n_out, F), n_out = at_phase_vocoder_frames(n_in, p, q) = ceil(n_in q / p)
 *   |Y_k| interpolates |X| linearly at t_k = k p / q; the phase advances by the wrapped
 *   inter-frame phase difference (the formulation of torchaudio.functional.phase_vocoder).
 */
int64_t at_phase_vocoder_frames(int64_t n_in, int64_t p, int64_t q);
int at_phase_vocoder_f32(const float* X, int64_t rows, int64_t n_in, int64_t F, int64_t p, int64_t q, int hop,
                         float* Y, int64_t n_out, void* stream);

/* ---- row peaks and impulse-response preparation -----------------------------------------------
 * Replaces whole-tensor torch chains of the reference by single passes:
 *   at_absmax_f32     audiotools/core/effects.py:100 (ir.abs().argmax), :118 (max|ir| clamp),
 *                     :160,175 (peak before / after apply_ir), :213-216 (ensure_max_of_audio)
 *                     vmax[r] = max_n |x[r,n]|, imax[r] = first n attaining it (imax may be NULL)
 *   at_roll_pad_f32   effects.py:85-100: `other` zero-padded / truncated to T and rotated so that
 *                     its |peak| sits at sample 0:  out[r,n] = xz[r, (n + shift[r]) mod T]
 *   at_alter_drr_f32  effects.py:540-647 decompose_ir + solve_alpha + alter_drr followed by
 *                     ensure_max_of_audio(1.0); x (B,C,T), drr (B) in dB, t0 = int(sr * 0.0025)
 */
int at_absmax_f32(const float* x, int64_t rows, int64_t T, float* vmax, int64_t* imax, void* stream);
int at_roll_pad_f32(const float* x, int64_t rows, int64_t L, const int64_t* shift, int64_t T, float* out, void* stream);

/* ---- windows along the batch axis -------------------------------------------------------------
 * Replaces  audiotools/core/dsp.py:70-108  collect_windows: torch unfold + permute + reshape of the (already padded) signal:
 *             out (rows * nw, win), nw = (T - win) / hop + 1,  out[(r nw + w), n] = x[r, w hop + n]
 *           audiotools/core/dsp.py:110-151 overlap_and_add: fold of the windows, fold of a tensor of ones, division and the
 *             trim of `trim` samples from either end, in one gather:  out (rows, out_len),
 *             out[r, t] = sum_w frames[r, w, t + trim - w hop] / #{windows covering t + trim}   (0 / 0 = NaN where none does,
 *             as the reference's folded / norm)
 */
/* ---- quantization ------------------------------------------------------------------------------
 * Replaces  audiotools/core/effects.py:452-486 (quantization) and :488-527 (mulaw_quantization): the chain of whole-tensor
 *           operations as one pass, one float32 operation per step of the chain and in its order.
 *   x, out (B, per_item) with per_item = C * T; q (B): quantization_channels (mulaw = 0) resp. quantization_channels - 1
 *   = mu (mulaw = 1) of every item, as float32
 */
int at_quantize_f32(const float* x, int64_t B, int64_t per_item, const float* q, int mulaw, float* out, void* stream);

int at_collect_windows_f32(const float* x, int64_t rows, int64_t T, int win, int hop, float* out, void* stream);
int at_overlap_add_f32(const float* frames, int64_t rows, int64_t nw, int win, int hop, int64_t trim, int64_t out_len, float* out,
                       void* stream);
int at_alter_drr_f32(const float* x, int64_t B, int64_t C, int64_t T, int t0, const float* drr, float* out,
                     void* stream);
/* the same, and in its output pass what at_absmax_f32(out) would report: vmax (B*C) and imax (B*C, may be NULL) -- the
 * peak and roll position effects.py:94-100, 118 take from the altered impulse response before the convolution */
int at_alter_drr_peak_f32(const float* x, int64_t B, int64_t C, int64_t T, int t0, const float* drr, float* out,
                          float* vmax, int64_t* imax, void* stream);

/* ---- polyphase resampling ---------------------------------------------------------------
 * Replaces  audiotools/core/audio_signal.py:732  julius.resample_frac(x, old, new) (zeros 24,
 *           rolloff 0.945): replicate pad (width, width+old), conv1d with the (new, 2*width+old)
 *           bank at stride old, interleave phases, cut to floor(new*T/old).
 *   wg (LG, NG, 4) f32 tap-major, base (NG) i32: the bank grouped 4 output phases per tap -- phases
 *          4G..4G+3 use dense taps base[G] .. base[G]+LG-1, zero filled outside each phase's own
 *          support, LG a multiple of 4 (tables.resample_grouped_bank); old_sr/new_sr are the REDUCED ratio.
 */
int at_resample_f32(const float* x, int64_t rows, int64_t T, const float* wg, const int* base, int old_sr, int new_sr,
                    int width, int NG, int LG, float* out, int64_t out_len, void* stream);

/* The same resampler on the matrix cores (v_mfma_f32_16x16x4_f32, exact f32): frames x phases x taps
 * as a banded GEMM, one LDS read of x per 16 FMAs instead of per 4.  Odd reduced `old_sr` only
 * (bank-conflict-free operand reads); at_resample_mfma_supported() tells.  W (NPB, NC, 2, 64, 4),
 * lo (NPB): tables.resample_mfma_bank; max_lo = max(lo).
 */
int at_resample_mfma_supported(int old_sr, int new_sr);
int at_resample_mfma_f32(const float* x, int64_t rows, int64_t T, const float* W, const int* lo, int old_sr, int new_sr,
                         int width, int NPB, int NC, int max_lo, float* out, int64_t out_len, void* stream);

/* The same sum on the fp16 matrix cores with float32-class accuracy (csrc/resample_f16.hip, round 4): samples (scaled
 * per 16-frame tile by a power of two) and taps (scaled per bank by 2^w_scale_log2) are split into fp16 high + low
 * halves and the three products hh + (hl + lh) are accumulated in fp32 by v_mfma_f32_16x16x32_f16 -- 16x the rate of
 * the f32 MFMA the entry point above is bound by.  2-4e-7 of the row maximum against float64 (the f32 kernels: 4-7e-7).
 * Odd reduced `old_sr`, 64 <= new_sr <= 256, T >= 16: at_resample_f16s_supported() tells; callers fall back to
 * at_resample_mfma_f32 / at_resample_f32 otherwise (AT_ERR_UNSUPPORTED).
 * W (NPB, NC, 2, 64, 4) uint32, lo (NPB), w_scale_log2: tables.resample_f16_bank; max_lo = max(lo).
 */
int at_resample_f16s_supported(int old_sr, int new_sr);
int at_resample_f16s_f32(const float* x, int64_t rows, int64_t T, const void* W, const int* lo, int old_sr, int new_sr,
                         int width, int NPB, int NC, int max_lo, int w_scale_log2, float* out, int64_t out_len, void* stream);

/* ---- circular FFT convolution -----------------------------------------------------------
 * Replaces  audiotools/core/effects.py:102-121  (rfft x 3, irfft x 2 at length T, rescale).
 *   x (B,C,T), ir (B,Cir,T) with Cir == 1 or C (already padded/rolled), scale (B,Cir) or NULL,
 *   out (B,C,T) = irfft(rfft(x) * rfft(ir)) * scale;  workspace: at_fftconv_workspace_bytes.
 */
int64_t at_fftconv_workspace_bytes(int64_t B, int64_t C, int64_t Cir, int64_t T);
int at_fftconv_circ_f32(const float* x, const float* ir, const float* scale, int64_t B, int64_t C, int64_t Cir,
                        int64_t T, float* out, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- the same convolution by a hand-written four-step FFT (no rocFFT) --------------------
 * Replaces  audiotools/core/effects.py:102-121  for the lengths it has a plan for: T even,
 * T/2 = N1 N2 with N1 <= 512, N2 <= 2048 and only the prime factors 2, 3, 5, 7 (every whole number
 * of seconds at 8 / 16 / 22.05 / 24 / 32 / 44.1 / 48 kHz up to ~40 s; at_longconv_supported tells).
 * Four launches: column FFTs of x and of the IR, one kernel that does the row FFTs, the real-FFT
 * split, the product, the inverse split and the row FFTs back, and the column FFTs back.
 *   tables: device copy of at_longconv_tables_host(T, buf, at_longconv_table_floats(T)) -- the
 *   twiddles, evaluated in double on the host; arguments otherwise as at_fftconv_circ_f32.
 *   x, ir and out must be 8-byte aligned; out doubles as the signal's work array.
 */
int at_longconv_supported(int64_t T);
int at_longconv_plan(int64_t T, int* n1, int* n2);
int64_t at_longconv_table_floats(int64_t T);
int at_longconv_tables_host(int64_t T, float* out, int64_t n);
int64_t at_longconv_workspace_bytes(int64_t B, int64_t C, int64_t Cir, int64_t T);
int at_longconv_circ_f32(const float* x, const float* ir, const float* scale, int64_t B, int64_t C, int64_t Cir,
                         int64_t T, const float* tables, float* out, void* workspace, int64_t workspace_bytes,
                         void* stream);
/* The same with the neighbouring passes of EffectMixin.apply_ir folded in.  Replaces
 * audiotools/core/effects.py:86-121 (zero padding of the IR to T, roll to its peak, convolution) and
 * the two abs().max() of :160 / :175:
 *   ir (B*Cir, ir_pitch) with ir_len <= T valid samples per row (the rest of the period counts as
 *   zero and is not read), ir_shift (B*Cir) int64 or NULL = rotation applied while reading
 *   (h[n] = ir[(n + shift) mod T]); x_peak / y_peak (B*C) or NULL receive max|x| and max|out| per
 *   row (NaN propagates), found while the column transforms stream the data.
 */
int at_longconv_room_f32(const float* x, const float* ir, int64_t ir_pitch, int64_t ir_len, const int64_t* ir_shift,
                         const float* scale, int64_t B, int64_t C, int64_t Cir, int64_t T, const float* tables, float* out,
                         float* x_peak, float* y_peak, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOTOOLS_AMD_H */
