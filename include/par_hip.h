/*
 * par_hip.h -- C ABI of libpar_hip.so, the MI355X (gfx950) implementation of the
 * pyaudiorestoration util/ hot path.  This is the drop-in boundary: plain pointers
 * and sizes, no torch / numpy / C++ types.  Every DEVICE pointer is caller-owned
 * HBM (the Python host layer allocates it through torch-ROCm tensors); `stream`
 * is a hipStream_t passed as void* (NULL = the default stream).  The library keeps
 * no global state except small per-device constant tables (twiddles, sinc tap
 * tables) keyed by (device, size).
 *
 * All functions return an int status (0 = PAR_OK), never throw, and record a
 * message retrievable with par_last_error() (thread-local).  Functions are
 * re-entrant and call hipSetDevice(device) themselves, so they may be driven from
 * one host thread per GPU (ctypes releases the GIL).
 *
 * Citations are file:line in the reference checkout (HENDRIX-ZT2/pyaudiorestoration).
 */
#ifndef PAR_HIP_H
#define PAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAR_OK 0
#define PAR_ERR_ARG 1            /* bad argument (the reference would raise or misbehave) */
#define PAR_ERR_HIP 2            /* a HIP runtime call failed */
#define PAR_ERR_UNSUPPORTED 3    /* valid for the reference, not implemented here (caller falls through) */
#define PAR_ERR_WORKSPACE 4      /* caller-provided workspace too small */
#define PAR_ERR_EMPTY_BAND 5     /* a tracker band is empty: the reference raises ValueError (argmax of an empty slice) */
#define PAR_ERR_INDEX 6          /* the reference indexes past the end of an array here and raises IndexError */
#define PAR_ERR_SHAPE 7          /* the reference multiplies arrays of different lengths here and raises ValueError */

/* ---- library / device ---------------------------------------------------------- */
int par_version(void);
int par_device_count(void);
int par_last_error(char* buf, int n);
/* stream-ordered helper used by the host layer and bench to time kernels with HIP events
 * on the stream the kernels are launched on (torch.cuda.Event only sees torch's stream). */
int par_event_create(void** ev);
int par_event_destroy(void* ev);
int par_event_record(void* ev, void* stream);
int par_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */
int par_stream_sync(int device, void* stream);
/* A side stream for the plan of the next file (resampling.varispeed_batch_dev, bench.py): cu_count > 0 confines its kernels
 * to the first cu_count compute units (hipExtStreamCreateWithCUMask), otherwise low_priority != 0 asks for the device's
 * least stream priority.  Wrap the handle with torch.cuda.ExternalStream; destroy it with par_stream_destroy. */
int par_stream_create(int device, int low_priority, int cu_count, void** stream);
int par_stream_destroy(void* stream);

/* ---- S0-S4: STFT / magnitude ---------------------------------------------------
 * Replaces the backend slot of util/fourier.py:67-75, i.e. a callable
 * (n_fft, step, window, x, zeropad) like pyfftw_rfft2 (:124-133) / np_rfft_pick (:136-157):
 * reflect-pad n_fft/2 (estimate_and_center :78-82), frame i = window * xpad[i*hop : i*hop+n_fft]
 * zero-extended at the END to n_fft*zeropad (segment_array :160-166), real FFT, / sqrt(n_fft)
 * (numpy/torch normalisation).  mode 1 fuses to_mag (:23-24): |X| + 1e-7.
 *
 *   x        device f32, logical length n, element stride x_stride (channel view of an (n,ch) array)
 *   window   device f32[n_fft] (scipy.signal.get_window(name, n_fft) as float32, :66)
 *   out      device, FRAME-MAJOR: mode 0 -> complex64 [frames][bins] interleaved re,im
 *                                 mode 1 -> float32   [frames][bins]
 *            bins = n_fft*zeropad/2+1, frames = par_stft_frames(n, n_fft, hop)
 * n_fft*zeropad must be a power of two in [16, 16384] (larger: par_stft_big_f32); otherwise PAR_ERR_UNSUPPORTED.
 */
int64_t par_stft_frames(int64_t n, int n_fft, int hop);
/* out_pitch (r03): elements (float, or float2 for mode 0) from one frame's row to the next; 0 = bins (packed rows).  A pitch
 * that is a multiple of 32 floats starts every magnitude row on a 128-byte line: a 513-bin row is 2052 bytes, packed rows
 * straddle lines at both ends and the streaming stores of neighbouring frames wrote 1.21x the bytes (WRITE_SIZE, r02). */
int par_stft_f32(int device, const float* x, int64_t n, int64_t x_stride, int n_fft, int hop, int zeropad,
                 const float* window, float* out, int mode, int64_t out_pitch, void* stream);

/* The same transform for frames of more than 8192 points (n_fft*zeropad a power of two in (8192, 2^24]; the GUI offers
 * FFT sizes up to 2^20 with zero-padding up to 16, util/widgets.py:333-351): up to 2^21 points a four-step FFT in two
 * passes over HBM (columns, then rows fused with the real-transform untangle: only one H-point array per frame is ever
 * stored); 2^22 .. 2^24 points (r03) as 2, 4 or 8 decimated 2^20-point sequences through that transform, recombined and
 * untangled bin by bin, a frame at a time (scratch: two H-point arrays).
 *   scratch  device memory of par_stft_big_scratch_bytes(n, n_fft, hop, zeropad) bytes (caller-owned) */
size_t par_stft_big_scratch_bytes(int64_t n, int n_fft, int hop, int zeropad);
int par_stft_big_f32(int device, const float* x, int64_t n, int64_t x_stride, int n_fft, int hop, int zeropad,
                     const float* window, float* out, int mode, void* scratch, size_t scratch_bytes, void* stream);

/* ---- S6: ISTFT -----------------------------------------------------------------
 * Replaces util/fourier.py:314-437 (istft, center=True, win_length=n_fft) incl.
 * window_sumsquare (:492-546) and __overlap_add (:677-687).  Does NOT mutate `spec`.
 *   spec     device complex64 [n_frames][bins] frame-major (bins = n_fft/2+1)
 *   window   device f32[n_fft]  (get_window(name, n_fft, fftbins=True))
 *   frames   device f32 scratch of par_istft_scratch_floats(n_frames, n_fft, hop) floats; that is 0 (pass NULL)
 *            when the frames are overlap-added in LDS and never stored (n_fft <= 2048 and the overlap-add span of
 *            one workgroup fits 64 KB), else [n_frames][n_fft] (n_fft >= 4096: a frame fills the workgroup alone and the
 *            fused form would re-transform n_fft/hop frames per hop of output); n_fft > 8192 (to 2^21: the GUI's FFT sizes go
 *            to 2^20, util/widgets.py:334-351): the frame array plus two complex arrays of the four-step transform
 *   y        device f32[y_len]; y[t] = ola[t + skip] / sumsq[t + skip] (0 beyond the ola length),
 *            skip = n_fft/2 and y_len = `length` reproduce fix_length(y[n_fft//2:], length) (:430-435).
 */
int64_t par_istft_scratch_floats(int64_t n_frames, int n_fft, int hop);
int par_istft_f32(int device, const float* spec, int64_t n_frames, int n_fft, int hop, const float* window,
                  float* frames, float* y, int64_t y_len, int64_t skip, void* stream);

/* Spectral gain mask of the dropout healer (dropout_healer_gui.py:161-162, util/units.py:28-29 to_fac):
 * spec[i] *= 10^(gain_db[i]/20) for a frame-major complex64 spectrogram and a float32 mask of `count` bins. */
int par_spec_apply_gain_db_c64(int device, float* spec, const float* gain_db, int64_t count, void* stream);

/* Sparse dropout healing (r03; dropout_healer_gui.py:111-166 touches only the marked boxes, and STFT -> ISTFT is the
 * identity elsewhere): copies `n_seg` sample ranges between two signals, dst[dst_start[k] + i] = src(src_start[k] + i)
 * for i < len[k].  run_start[k] = sum of len[0..k) (exclusive prefix), total = sum of len.  padded != 0: the source is the
 * reference's fix_length(signal, n_padded) (zeros from sample n_valid on, :120) under the STFT's np.pad(.., 'reflect')
 * (util/fourier.py:78-82), so a range may start before sample 0 or end behind n_padded.  All index arrays on the device. */
int par_copy_segments_f32(int device, const float* src, int64_t src_stride, int64_t n_valid, int64_t n_padded, int padded,
                          const int64_t* src_start, const int64_t* dst_start, const int64_t* len, const int64_t* run_start,
                          int64_t n_seg, int64_t total, float* dst, int64_t dst_stride, void* stream);
/* Gain mask of the dropout healer for a BATCH of markers (dropout_healer_gui.py:135-159): for each marker
 * (frame_b, frame_a, fs, bin_l, bin_u -- int32[n_markers][5] in device memory, the integers the reference
 * derives at :136-142) the mean dB of the fs frames before/after the box per bin, the linear fill across the
 * box (:149-153), gain = target - dB (:155) and np.clip(gain, previous, 255) (:157) folded into the mask.
 *   spec     device complex64 [n_frames][bins], frame-major (par_stft_f32 output); read only
 *   gain_db  device f32 [n_frames][bins], zero-initialised by the caller; updated with an atomic max, which
 *            equals the reference's marker-by-marker clip because the mask is non-negative (heal.hip header)
 * A marker whose box or surrounding frames leave the spectrogram contributes nothing. */
int par_inpaint_gain_db_c64(int device, const float* spec, int64_t n_frames, int64_t bins, const int32_t* markers,
                            int64_t n_markers, float* gain_db, void* stream);

/* Apply the mask over the marker boxes only and clear it (dropout_healer_gui.py:161-162 restricted to where the
 * mask can be non-zero): spec[i] *= 10^(gain_db[i]/20), gain_db[i] = 0 for every bin of every box; a bin inside
 * several boxes is scaled once.  `markers` as for par_inpaint_gain_db_c64.  Leaves gain_db all zeros, ready for
 * the next file, so a batch driver zero-fills the mask once per allocation, not once per file. */
int par_spec_apply_gain_boxes_c64(int device, float* spec, int64_t n_frames, int64_t bins, const int32_t* markers,
                                  int64_t n_markers, float* gain_db, void* stream);

/* Band volume curve of the dropout detector (dropout_healer_gui.py:195-203):
 * out[i] = mean_b 20*log10(mag[frame_b+i][b]), b in [bin_l, bin_u), i in [0, frame_a-frame_b); f64 out.
 *   mag  device f32 [n_frames][bins] frame-major magnitude (get_mag output, already + 1e-7). */
int par_band_mean_db_f32(int device, const float* mag, int64_t n_frames, int64_t bins, int64_t mag_pitch, int bin_l, int bin_u,
                         int64_t frame_b, int64_t frame_a, double* out, void* stream);

/* Heuristic dropout repair (dropouts_gui.MainWindow.process_heuristic, dropouts_gui.py:241-323): the two O(n) passes around
 * the band-pass of a band, over the n_ch channels of an interleaved (n, sig_stride) float32 file at once.
 *   par_curve_scale_f64:    out[c][i] = sig[i][c] * np.interp(np.linspace(0, 1, n)[i], np.linspace(0, 1, frames), fac[c])  (:314-317;
 *                           fac device f64[n_ch][frames] = correction factor - 1, out device f64[n_ch][n])
 *   par_accumulate_f64_f32: sig[i][c] = float32(float64(sig[i][c]) + y[c][i])                                                (:319-321) */
int par_curve_scale_f64(int device, const float* sig, int64_t sig_stride, int n_ch, int64_t n, const double* fac, int64_t frames,
                        double* out, void* stream);
int par_accumulate_f64_f32(int device, float* sig, int64_t sig_stride, int n_ch, int64_t n, const double* y, void* stream);

/* ---- R1: speed curve -> fractional read positions ----------------------------------
 * Replaces resampling.speed_to_pos (util/resampling.py:93-137).  Two calls because the
 * caller must allocate the position array:
 *   plan : segment lengths n_i (error-diffused rounding :111-118), per-segment reciprocal sums,
 *          the float64 offset chain (:125-126) and the end-trim (:129-135).  Returns len_out
 *          (= the written prefix when no trim fires) and whether the trim fired.
 *          Synchronises `stream`.
 *   fill : writes pos[0..len_out) as float64, bit-identical to the reference's array.
 * sampletimes/speeds are DEVICE float64[m].  work is caller-owned device scratch of at least
 * par_speed_plan_bytes(m) bytes and must stay untouched between plan and fill (or the fused
 * resampler below).
 * Curves with FEW points (segments of thousands of samples and more -- a constant speed correction is two points):
 * use par_speed_to_pos_plan_fused + par_speed_to_pos_fill_fused.  This plain form walks every segment on one lane
 * (seconds for a 10^8-sample segment); the fused plan's checkpoint buffer is what lets long segments be cut into
 * chunks whose float64 cumsum chain is evaluated exactly in parallel.
 */
size_t par_speed_plan_bytes(int64_t m);
int par_speed_to_pos_plan(int device, const double* sampletimes, const double* speeds, int64_t m, int64_t n_in,
                          void* work, size_t work_bytes, int64_t* len_out, int* trimmed, void* stream);
/* Same as par_speed_to_pos_plan; force_host != 0 runs the serial host evaluation of the two
 * order-dependent chains (the exact fallback the device scans defer to when they flag a near-tie),
 * *path_used (optional) reports 0 = device scans, 1 = serial host path, 2 = segment lengths redone on the host after a
 * near-tie (O(m)) with everything else on the device.  force_host == 2 (fused form, tests only) additionally injects
 * the failure the exact chunked cumsum reports when its verification does not close: the plan stays valid, *fused_ok
 * must come back 0. */
int par_speed_to_pos_plan_ex(int device, const double* sampletimes, const double* speeds, int64_t m, int64_t n_in,
                             void* work, size_t work_bytes, int64_t* len_out, int* trimmed, int force_host,
                             int* path_used, void* stream);
/* Why this thread's most recent plan left the device path: the device scans' flag word (1 near-tie in the segment
 * lengths, 2 some n_i < 2, 4 a_i outside the exact fixed-point range, 8 offset-chain verification failed, 16 more
 * binade crossings than the stitch table holds, 64 end_guess within the device sum's error of an integer); 0 when the
 * plan ran on the device (path_used 0) or the host path was forced.  Diagnostics only. */
int par_last_plan_flags(void);
int par_speed_to_pos_fill(int device, const double* speeds, int64_t m, const void* work,
                          double* pos, int64_t len_out, void* stream);
/* The same fill from a FUSED plan (fused_ok): parallel over 8-sample blocks restarting from the cumsum checkpoints,
 * so a curve with a handful of points (segments of 10^7..10^9 samples) fills in milliseconds; bit-identical. */
int par_speed_to_pos_fill_fused(int device, const double* speeds, int64_t m, const void* work, const void* aux,
                                int64_t max_out, double* pos, int64_t len_out, void* stream);

/* ---- R2/R3: windowed-sinc varispeed interpolation --------------------------------------
 * Replaces sinc_wrapper / sinc_wrapper_mt / sinc_core (util/resampling.py:21-90), operator
 * slot #2: out[i] = sum_k signal[lower+k] * sinc((N_k-shift)*fc)*fc * hanning(2NT+1)[k].
 * Canonical period definition: period_to[i] = max(1e-12, pos[i+1]-pos[i])
 * for all but the global last sample (which reuses the previous one) == single-thread
 * sinc_wrapper.  Bug-compatible leading edge (quirk 1).  len_out >= 2, 1 <= NT <= 512.
 *   pos   device f64[len_out];  sig device f32 (len_in, stride sig_stride);
 *   out   device f32 (len_out, stride out_stride)  -- strided column views of (n,ch) arrays.
 * Positions may be ANY finite float64 (non-monotonic, repeated, negative, past the end, beyond the int64 range):
 * the window is clipped like the reference's slice signal[max(0, ind-NT) : min(ind+NT, len)], an empty slice
 * sums to 0.  Non-finite positions make the reference raise (int(round(p))); the Python mirror checks for them,
 * this entry point leaves the corresponding outputs unspecified (0 or NaN).
 */
int par_sinc_resample_f32(int device, const double* pos, int64_t len_out, const float* sig, int64_t sig_stride,
                          int64_t len_in, int NT, float* out, int64_t out_stride, void* stream);

/* Higher-level slot of resampling.run (util/resampling.py:184, :225-227): positions + sinc interpolation of
 * one channel from a finished plan.  n_chunks > 1 pipelines the work in pieces (position fill of chunk c+1
 * on a library-owned side stream while chunk c is interpolated); 0 = auto (= 1 on MI355X, where the overlap
 * measured no gain).  Bit-identical to par_speed_to_pos_fill + par_sinc_resample_f32.  pos is caller-owned f64[len_out]
 * scratch (it holds the reference's sample_at array afterwards). */
int par_varispeed_resample_f32(int device, const double* speeds, int64_t m, const void* work, int64_t len_out,
                               double* pos, const float* sig, int64_t sig_stride, int64_t len_in, int NT, float* out,
                               int64_t out_stride, int n_chunks, void* stream);
/* Fused resampler: positions never touch HBM.  par_speed_to_pos_plan_fused is par_speed_to_pos_plan_ex plus, in the
 * caller-owned device buffer `aux` (>= par_fused_aux_bytes(max_out, m) bytes, max_out >= len_out, e.g. 1.02*n_in), a
 * checkpoint of every segment's running reciprocal sum every 8 steps and a tile -> segment map.
 * par_varispeed_fused_f32 then runs K_sinc with each workgroup regenerating its tile's float64 positions in LDS
 * from those checkpoints (sequential float64 adds, bit-identical to numpy's cumsum) -- same output as
 * par_speed_to_pos_fill + par_sinc_resample_f32, 16 B/sample less HBM traffic.  *fused_ok (optional) reports
 * whether the checkpoints of every needed segment fit in aux; only then may par_varispeed_fused_f32 be called
 * (with the same work/aux/max_out and the returned len_out -- it does not read the plan back, so that it never
 * synchronises); otherwise callers use the position-array path (the plan itself stays valid).
 * LAZY plans (r05, csrc/pos_plan.h): on a dense, gentle curve (every segment 2..1024 outputs, speeds in [1/16, 64] changing
 * by <= 2^-9 of themselves across it) the plan skips the per-sample cumsum: segment sums in closed form with a rigorous error
 * bound, the exact sequential sum only for the few thousand segments where that bound could change a rounding of the offset
 * chain, and K_sinc walks a segment's cumsum with the reference's own arithmetic only for the ~1 output in 10^6 whose
 * position lies within the bound of a half-integer.  Offsets, len_out, trim and every window centre are bit-identical to
 * the eager plan's; *fused_ok comes back 2 and aux then holds NO checkpoints (par_speed_to_pos_fill_fused must not be used).
 * Curves that do not qualify are planned the eager way (*fused_ok 1).  force_host | 8 asks for the eager plan outright. */
size_t par_fused_aux_bytes(int64_t max_out, int64_t m);
int par_speed_to_pos_plan_fused(int device, const double* sampletimes, const double* speeds, int64_t m, int64_t n_in,
                                void* work, size_t work_bytes, void* aux, size_t aux_bytes, int64_t max_out,
                                int64_t* len_out, int* trimmed, int force_host, int* path_used, int* fused_ok,
                                void* stream);
int par_varispeed_fused_f32(int device, const double* speeds, int64_t m, const void* work, const void* aux,
                            int64_t max_out, int64_t len_out, const float* sig, int64_t sig_stride, int64_t len_in,
                            int NT, float* out, int64_t out_stride, void* stream);

/* Which kernel runs: NT = 32 files of at least four 1024-output tiles and 4096 input samples -- mono on unit strides, ONE
 * channel of a two-channel interleaved file (sig_stride == 2, any out_stride: the reference's use_channels column views,
 * util/resampling.py:211-227; r06), or (par_varispeed_fused_stereo_f32) the two channels of an INTERLEAVED file (sig1 = sig0 + 1,
 * out1 = out0 + 1, strides 2, out0 8-byte aligned) -- take the streaming kernel (csrc/sinc2.hip: one wave per 9-23 tiles, the taps |n| >= 3 of both tap regimes
 * as fixed filter banks on the matrix cores, fc < 1 through seven moment filters; stereo: one placement for both channels).  The
 * file's end tiles (first, last two, the partial one) are done the block kernel's way by the first workgroups of the same
 * launch; tiles the streams do not cover -- blocks outside the record model, window-centre ties, input float16 does not suit --
 * go to the block kernel (csrc/sinc.hip, sinc_block.h) through a tile list in `aux`.  Everything else (other NT, channels of
 * files with three or more channels, planar channel pairs, short files) takes the block kernel.  Results agree within the contract's tolerance and every window
 * centre is the reference's either way.  K_sinc WRITES that list into `aux`: one par_varispeed_fused_* launch per plan at a time
 * (two launches of one plan on different streams would race on it). */

/* Process-wide choice of that kernel, for tests and A/B sessions: form -1 = the default above, 0 = the block kernel for
 * everything.  Returns the previous setting. */
int par_debug_sinc_kernel(int form);

/* Diagnostic: how many 1024-output tiles of the LAST par_varispeed_fused_* launch on this aux buffer the streaming kernel
 * handed to the block kernel's tile list (blocks outside the record model, window-centre ties, input that float16 does not
 * suit).  Synchronises the stream.  0 when the streaming kernel did not run (the plan zeroes the count). */
int par_fused_redo_tiles(int device, const void* aux, int64_t max_out, int64_t m, int* tiles, void* stream);
/* ... and which tiles: *count = the length of that list, tiles[0 .. min(*count, cap)) = its entries (host buffer; the order is
 * the order in which the streams pushed them).  Tests aim their oracle windows at them.  Synchronises the stream. (ABI 105) */
int par_fused_redo_list(int device, const void* aux, int64_t max_out, int64_t m, int* tiles, int cap, int* count, void* stream);

/* Several planned files in ONE call (ABI 106; the reference loops over files, util/resampling.py:168, and an archive of short files
 * pays per file for the tails and gaps of the two or three kernels a file's K_sinc is): items that ALL take the streaming kernel in
 * one form -- NT = 32 and every item mono on unit strides, or every item an interleaved stereo file (sig1 = sig0 + 1, out1 = out0 + 1,
 * strides 2, out0 8-byte aligned) -- are launched merged, up to eight files per launch; any other mix is done item by item.
 * Outputs are bit-identical to par_varispeed_fused_f32 / par_varispeed_fused_stereo_f32 called per item (each file's streams are cut
 * as in its own launch).  sig1 == out1 == NULL: one channel.  Every item needs its own plan buffers (work, aux). */
typedef struct par_fused_item {
  const double* speeds;
  int64_t m;
  const void* work;
  const void* aux;
  int64_t max_out, len_out;
  const float* sig0;
  const float* sig1;
  int64_t sig_stride, len_in;
  float* out0;
  float* out1;
  int64_t out_stride;
} par_fused_item;
int par_varispeed_fused_batch_f32(int device, int n_items, const par_fused_item* items, int NT, void* stream);

/* Stereo form: two channels of ONE file (same positions; sig0/sig1 and out0/out1 share the strides -- e.g. the two
 * columns of an interleaved (n, 2) array: sig1 = sig0 + 1, stride 2) in one launch.  Outputs equal two
 * par_varispeed_fused_f32 calls to float32 rounding (the lane/output map differs); position regeneration, prologue and tap
 * weights are evaluated once for both.  Interleaved NT = 32 files take the streaming kernel's stereo form (see above). */
int par_varispeed_fused_stereo_f32(int device, const double* speeds, int64_t m, const void* work, const void* aux,
                                   int64_t max_out, int64_t len_out, const float* sig0, const float* sig1,
                                   int64_t sig_stride, int64_t len_in, int NT, float* out0, float* out1,
                                   int64_t out_stride, void* stream);

/* Profiling hook (bench.py roofline leg): HIP-event timing, on the caller's stream, of the K_sinc launches
 * issued by the last par_varispeed_resample_f32 call on `device`. */
int par_profile_enable(int device, int on);
int par_profile_read(int device, float* total_ms, int* launches, int64_t* samples);

/* "Linear" mode of resampling.run (util/resampling.py:229): np.interp(pos, arange(len), sig, 0, 0). */
int par_linear_resample_f32(int device, const double* pos, int64_t len_out, const float* sig, int64_t sig_stride,
                            int64_t len_in, float* out, int64_t out_stride, void* stream);

/* Lag-curve branch of resampling.run (util/resampling.py:189-206): pos = clip(np.interp(arange(num_out), xp, fp), 0),
 * cut at the first value >= len_signal (find_cutoff :265-270).  np.interp's operation order is kept (no FMA),
 * so positions are bit-identical.
 *   xp, fp   device f64[m], m >= 2: sampletimes and sampletimes - lags (samples)
 *   pos      device f64[num_out] (caller-allocated; only the first *len_out entries are meaningful)
 *   work     device scratch, >= 8 bytes
 *   len_out  host: num_out, or the cut-off index; *trimmed = 1 when the cut-off fired.  Synchronises the stream. */
int par_lag_to_pos_f64(int device, const double* xp, const double* fp, int64_t m, int64_t num_out, int64_t len_signal,
                       double* pos, void* work, int64_t* len_out, int* trimmed, void* stream);

/* ---- host-side codec: FLAC decoder standing in for soundfile/libsndfile (util/io_ops.py:7-16) ----
 * Pure host code, no GPU needed.  `data` is the whole file in host memory.
 *   par_flac_info        STREAMINFO fields; md5 (16 bytes, optional) is the stream's PCM signature
 *   par_flac_decode_f32  out: host float32 [total_frames][channels], scaled by 2^-(bits-1) like libsndfile's float
 *                        read; frame-parallel over n_threads (<= 0: all cores); every frame is CRC-checked;
 *                        verify_md5 != 0 also checks the decoded PCM against STREAMINFO's MD5. */
int par_flac_info(const void* data, size_t nbytes, int* sample_rate, int* channels, int* bits, int64_t* total_frames,
                  uint8_t* md5);
int par_flac_decode_f32(const void* data, size_t nbytes, float* out, int64_t frames_cap, int n_threads, int verify_md5,
                        int64_t* frames_decoded);

/* ---- synthetic workload generators, so bench inputs are born in HBM ---------- */
int par_synth_signal_f32(int device, float* out, int64_t start, int64_t count, double sr, uint64_t seed, void* stream);
int par_synth_speed_curve_f64(int device, double* sampletimes, double* speeds, int64_t m, double duration_s,
                              double sr, double depth, double rate_hz, double phase, void* stream);

/* ---- W1/W2: per-frame peak trackers on a device magnitude spectrogram ---------------------
 * Replaces the trace() loops of PeakTracker / PeakTrackTracker (util/wow_detection.py:294-327)
 * with Track.set_bin_limits (:97-107), get_peak (:119-134), is_peak (:136-139) and
 * correlation.parabolic (util/correlation.py:42-46).
 *   mag      device f32 frame-major [n_frames][bins], rows mag_pitch floats apart (0 = bins: packed; par_stft_f32's
 *            out_pitch).  The same parameter on par_track_cog_f64 / par_track_corr_f64 / par_piptrack_f32 /
 *            par_band_mean_db_f32.
 *   freqs    device f64[count]: in = sampled trail (sample_trail :66-76), out = traced frequencies
 *   mode 0   PeakTracker: band re-centred on freqs[i] every frame
 *   mode 1   PeakTrackTracker: band fixed on freqs[0]; tolerance halves for i > 2
 *   status   device int32 scratch (4 bytes).  A band whose widening reaches below bin 0 is an empty slice in the
 *            reference (its argmax raises ValueError): reported as PAR_ERR_EMPTY_BAND, never clamped; a peak on the last
 *            bin is its IndexError (is_peak reads the bin above): PAR_ERR_INDEX; a Center-of-Gravity band past the last
 *            bin is its broadcast ValueError: PAR_ERR_SHAPE.  Synchronises.
 */
int par_track_peak_f64(int device, const float* mag, int64_t n_frames, int bins, int64_t mag_pitch, int64_t frame_0,
                       int64_t count, double* freqs, int fft_size, double sr, double tolerance_oct, int mode, int32_t* status,
                       void* stream);
/* The same two trackers with the band magnitudes RE-EVALUATED FROM THE SIGNAL in float64 (r03): the reference's numpy
 * backend (util/fourier.py:136-157) hands Track.get_peak (util/wow_detection.py:119-134) float64 containers, and the
 * config-3 chain amplifies the float32 noise of a spectrogram into 3.7e-5 of the output peak.  Per frame only the band
 * [NL, NU) and the two neighbours of its peak are needed: a windowed direct DFT of the reference's own float32 frame
 * (reflect pad n_fft/2, sample x window rounded to float32: segment_array :160-166; zero-extended to n_fft * zeropad;
 * / sqrt(n_fft); + 1e-7 as to_mag :23-29 adds).
 *   x        device f32 signal channel, element stride x_stride, n samples;  window  device f32[n_fft]
 *   bins = n_fft * zeropad / 2 + 1, n_frames <= n / hop + 1; freqs / mode / status as par_track_peak_f64.
 * A band wider than 2048 bins: PAR_ERR_UNSUPPORTED (use par_track_peak_f64 on the spectrogram).  Synchronises. */
int par_track_peak_refined_f64(int device, const float* x, int64_t n, int64_t x_stride, int n_fft, int hop, int zeropad,
                               const float* window, int bins, int64_t n_frames, int64_t frame_0, int64_t count, double* freqs,
                               double sr, double tolerance_oct, int mode, int32_t* status, void* stream);
/* CenterOfGravity.trace (util/wow_detection.py:256-291): sequential band adaptation. */
int par_track_cog_f64(int device, const float* mag, int64_t n_frames, int bins, int64_t mag_pitch, int64_t frame_0, int64_t count,
                      double* freqs, int fft_size, double sr, double tolerance_oct, int32_t* status, void* stream);

/* W4: CorrelationTracker.trace (util/wow_detection.py:396-436) for all frames in one call: the band [NL, NU) of frames
 * 0 .. count-1 (the reference ignores frame_0 here) is resampled onto n = 4 (NU - NL) log2-frequency points by the
 * quadratic spline of scipy.interpolate.interp1d(kind='quadratic'), Hann-windowed, cross-correlated frame against next
 * frame (util/correlation.py:6-13, mode 'same'; the frame behind the last is all ones), peak refined by parabolic()
 * (:42-46), changes summed in order.
 *   M        device f64 [n][NU-NL] row-major: the spline as a matrix (band values -> grid values; the spline is linear
 *            in the data, the host applies scipy's own construction to the identity)
 *   wind     device f64 [n] = np.hanning(n)
 *   log_span = log2 f[NU-1] - log2 f[NL], log_mean = log2((fL + fU)/2): the reference's final scaling
 *   work     device f64 [par_track_corr_work_len(count, n)];  freqs  device f64 [count] out
 *   status   device int32 scratch.  A peak on the last lag is the reference's IndexError: PAR_ERR_INDEX.  Synchronises. */
/* PartialsTracker's librosa.piptrack (util/wow_detection.py:361-387; third-party routine, restated from librosa 0.10's
 * published source, parity unpinned): thresholded local maxima of every magnitude column inside [fmin, fmax), refined by
 * a parabola.  mag: device f32 [n_frames][bins] as par_stft_f32 mode 1 writes it; S = (mag - offset) * scale gives the
 * |stft| librosa works on (offset 1e-7, scale sqrt(n_fft)).  pitches / mags: device f32 [n_frames][bins] out. */
int par_piptrack_f32(int device, const float* mag, int64_t n_frames, int bins, int64_t mag_pitch, float scale, float offset, int fft_size,
                     double sr, double fmin, double fmax, float threshold, float* pitches, float* mags, void* stream);
int64_t par_track_corr_work_len(int64_t count, int n);
int par_track_corr_f64(int device, const float* mag, int64_t n_frames, int bins, int64_t mag_pitch, int NL, int NU, int64_t count,
                       const double* M, const double* wind, int n, double log_span, double log_mean, double* work,
                       double* freqs, int32_t* status, void* stream);

/* X2: util/correlation.py:6-39 on the device.  a, b: float64 device signals (what butter_bandpass_filter hands to
 * find_delay, pytapesynch_gui.py:128-131).
 *   par_xcorr_f64       scipy.signal.correlate(a/|a|, b/|b|, 'full'): full[na + nb - 1] float64, through one complex
 *                       float32 transform of a + i b (values carry ~1e-6 of the peak)
 *   par_find_delay_f64  find_delay(a, b, ignore_phase) with the window already applied by the caller (the reference
 *                       multiplies a and b in place): the peak of the 'same' correlation is located by the transform,
 *                       the lags around it are re-evaluated as float64 dot products and parabolic() (:42-46) runs on
 *                       those; *delay = peak - na//2 (host), *corr = its height.  A peak on the last lag is the
 *                       reference's IndexError: PAR_ERR_INDEX.  Synchronises.
 *   scratch             device bytes of par_xcorr_scratch_bytes(na, nb).  na + nb - 1 <= 2^20: one transform; longer signals
 *                       (up to 2^25 samples each, else PAR_ERR_UNSUPPORTED) are correlated in pairs of 2^19-sample sections */
size_t par_xcorr_scratch_bytes(int64_t na, int64_t nb);
int par_xcorr_f64(int device, const double* a, int64_t na, const double* b, int64_t nb, void* scratch, size_t scratch_bytes,
                  double* full, void* stream);
int par_find_delay_f64(int device, const double* a, int64_t na, const double* b, int64_t nb, int ignore_phase, void* scratch,
                       size_t scratch_bytes, double* delay, double* corr, void* stream);

/* W3: sign-change indices of a (band-passed) float64 signal, ascending -- zero_crossings(a) =
 * np.where(np.bitwise_xor(a[1:] > 0, a[:-1] > 0))[0] (util/wow_detection.py:448-450), the full-rate pass of
 * ZeroCrossingTracker.trace (:340).
 *   work   device int64[par_zero_crossings_work_len(n)]
 *   idx    device int64[cap], or NULL to only count; *count (host) = number of crossings.  Synchronises the stream. */
int64_t par_zero_crossings_work_len(int64_t n);
int par_zero_crossings_f64(int device, const double* x, int64_t n, int64_t* work, int64_t* idx, int64_t cap, int64_t* count,
                           void* stream);

/* ---- F1: zero-phase SOS filtering ------------------------------------------------------
 * Replaces scipy.signal.sosfiltfilt(sos, data) as called by butter_bandpass_filter
 * (util/filters.py:24): odd extension of padlen samples, sosfilt_zi initial state scaled by the
 * first sample, forward pass, backward pass, trim.  The filter DESIGN (scipy.signal.butter,
 * sosfilt_zi; O(order) work) stays in the host layer; the O(n) recurrences run on the GPU.
 *   sos HOST f64[n_sections][6] (a0 == 1), zi HOST f64[n_sections][2] (= scipy.signal.sosfilt_zi(sos))
 *   x, y DEVICE f64[n];  work DEVICE f64[work_len], work_len >= par_sosfiltfilt_work_len(n, padlen)
 *   padlen = 3*(2*n_sections+1 - min(#(sos[:,2]==0), #(sos[:,5]==0)))  (scipy default), n > padlen.
 */
int64_t par_sosfiltfilt_work_len(int64_t n, int64_t padlen);
int par_sosfiltfilt_f64(int device, const double* sos, const double* zi, int n_sections, const double* x, int64_t n,
                        int64_t padlen, double* work, int64_t work_len, double* y, void* stream);

/* Batched form (r05): n_sig signals of ONE length n (signal i at x + i * x_stride, result at y + i * y_stride), n_filt = 1
 * (one cascade for all) or n_sig (cascade i for signal i: sos HOST f64[n_filt][n_sections][6], zi HOST f64[n_filt][n_sections][2]),
 * every stage ONE launch over the whole batch -- what dropouts_gui.process_heuristic does per band over the channels of a file
 * (dropouts_gui.py:314-321), or a multi-band analysis of one signal.  Each signal's result equals par_sosfiltfilt_f64's bit for
 * bit (same block / super-block chain).  padlen as above (the cascades of one call share it); n_sig <= 65535.
 * Synchronises the stream once (the parameter table's upload). */
int64_t par_sosfiltfilt_batch_work_len(int64_t n, int64_t padlen, int n_sig, int n_sections);
int par_sosfiltfilt_batch_f64(int device, const double* sos, const double* zi, int n_filt, int n_sections, const double* x,
                              int64_t x_stride, int n_sig, int64_t n, int64_t padlen, double* work, int64_t work_len, double* y,
                              int64_t y_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PAR_HIP_H */
