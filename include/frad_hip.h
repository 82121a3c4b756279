/*
 * frad_hip.h -- C-ABI of the MI355X-native FrAD transform core (libfrad_hip.so).
 *
 * Drop-in boundary for ONE hot path of H4n-uL/FrAD_Python: the per-frame Fourier
 * analysis/synthesis + quantise + bit-depth pack/unpack that the reference reaches only through
 * the two `match` dispatches
 *      src/libfrad/encoder.py:96-100   fourier.profileN.analogue(frame, bits, srate, endian|loss_level)
 *      src/libfrad/decoder.py:70-74    fourier.profileN.digital(frad, depth_idx, channels, ...)
 * Each entry point below replaces one of those Python functions with ONE batched launch over
 * `n_frames` independent frames of identical geometry.  The reference has no FFI of its own
 * (it is pure Python); INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) owned by the caller; nothing is allocated per call
 *    except the immutable twiddle tables, created once per (N, precision) -- call
 *    frad_plan_prepare() first if the launch must be hipGraph-capturable;
 *  - all work is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *  - return value: FRAD_OK (0) or a negative frad_status; never exit(), never throws;
 *  - "sample-frame" = one sample of every channel; PCM is interleaved [n, C] exactly as the
 *    reference reshapes it (encoder.py:84);
 *  - there is NO CPU fallback: without a HIP device every launch returns FRAD_E_HIP.
 */
#ifndef FRAD_HIP_H
#define FRAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRAD_ABI_VERSION 1

typedef enum frad_status {
    FRAD_OK = 0,
    FRAD_E_INVALID = -1,      /* bad argument (null pointer, illegal depth, N or C out of range)   */
    FRAD_E_UNSUPPORTED = -2,  /* legal in the reference but not built here yet (see strerror)      */
    FRAD_E_HIP = -3,          /* a HIP runtime call failed; frad_last_hip_error() has the code     */
    FRAD_E_NOMEM = -4
} frad_status;

/* PCM element type: kind*8 + log2(itemsize)*2 + big_endian, kind 0 = unsigned int, 1 = signed
 * int, 2 = IEEE float.  Replaces ff_format_to_numpy_type (src/libfrad/backend/pcmformat.py:4-32). */
#define FRAD_PCM_CODE(kind, log2size, be) ((kind) * 8 + (log2size) * 2 + (be))
enum {
    FRAD_PCM_U8 = 0, FRAD_PCM_U16LE = 2, FRAD_PCM_U16BE = 3, FRAD_PCM_U32LE = 4, FRAD_PCM_U32BE = 5,
    FRAD_PCM_U64LE = 6, FRAD_PCM_U64BE = 7,
    FRAD_PCM_S8 = 8, FRAD_PCM_S16LE = 10, FRAD_PCM_S16BE = 11, FRAD_PCM_S32LE = 12, FRAD_PCM_S32BE = 13,
    FRAD_PCM_S64LE = 14, FRAD_PCM_S64BE = 15,
    FRAD_PCM_F16LE = 18, FRAD_PCM_F16BE = 19, FRAD_PCM_F32LE = 20, FRAD_PCM_F32BE = 21,
    FRAD_PCM_F64LE = 22, FRAD_PCM_F64BE = 23
};

/* flags */
#define FRAD_LITTLE_ENDIAN 1u /* payload endianness (ASFH `endian`, tools/asfh.py:6-10); 12-bit is always BE */
#define FRAD_RAW_BE_INTS   2u /* reproduce the reference's to_f64 quirk: big-endian integer PCM is NOT
                                 normalised (pcmformat.py:37-45 compares against native dtypes only)  */

/* ---- tables / bookkeeping ------------------------------------------------------------------ */
int         frad_abi_version(void);
const char* frad_strerror(int status);
int         frad_last_hip_error(void);
/* payload bytes of one frame: N*C*bits/8, 12-bit rounds the nibble count up (profile0.py:33-41). */
size_t      frad_payload_bytes(int32_t N, int32_t C, int32_t bits);
/* 1 if frad_p0_* runs the LDS-resident FFT kernel for this N (else the generic direct-sum kernel). */
int         frad_has_fast_path(int32_t N, int32_t C, int32_t compute_f32);
/* build the twiddle tables for (N, precision) on the current device now (hipMalloc + upload).   */
int         frad_plan_prepare(int32_t N, int32_t compute_f32);
void        frad_plan_clear(void);

/* ---- profile 0: DCT archiving ----------------------------------------------------------------
 * frad_p0_analogue  == fourier.profile0.analogue (src/libfrad/fourier/profile0.py:14-44) fused with
 * to_f64 (backend/pcmformat.py:34-47) and the frame cut (encoder.py:72-93), batched.
 *   pcm            first sample-frame of frame 0; frame i starts `frame_stride` sample-frames later
 *   pcm_dtype      FRAD_PCM_*; integer and f64 input is transformed in f64, f32/f16 input in f32
 *                  (the reference does not widen floats and pocketfft then stays in single)
 *   bits           storage depth 12/16/24/32/48/64 (the caller applies `bits not in DEPTHS -> 16`)
 *   payload        frame i's bytes start at payload + i*payload_stride (>= frad_payload_bytes)
 *   absmax[i]      max |X| of frame i as float64 (NaN if any bin is NaN, like np.max): the caller
 *                  compares it with the storage type's max and re-submits the rare frame that
 *                  needs a deeper format (profile0.py:24-26).  May be NULL.
 */
int frad_p0_analogue(const void* pcm, int32_t pcm_dtype, int64_t n_frames, int32_t N, int32_t C,
                     int64_t frame_stride, int32_t bits, uint32_t flags,
                     void* payload, int64_t payload_stride, double* absmax, void* stream);

/* frad_p0_analogue with the reference's per-frame overflow test (profile0.py:24-26) applied in the same pass:
 * *overflow_flag is set to 1 when any frame's max|X| exceeds the storage float's largest finite value (NaN does not
 * count, as in the reference), and left alone otherwise.  The N = 2048 wave kernels test as they go; every other
 * geometry runs frad_p0_overflow_scan behind the transform on the same stream.  `absmax` is required.              */
int frad_p0_analogue_checked(const void* pcm, int32_t pcm_dtype, int64_t n_frames, int32_t N, int32_t C,
                             int64_t frame_stride, int32_t bits, uint32_t flags, void* payload, int64_t payload_stride,
                             double* absmax, int32_t* overflow_flag, void* stream);
/* The overflow test of profile0.py:24-26 over a batch, on the device: *flag |= 1 if any absmax[i] is
 * greater than the largest finite value of the `bits` storage float (NaN never is, like numpy's
 * comparison).  `flag` is a device int32 the caller zeroes once and reads when it needs the answer
 * (sticky across calls), so a steady stream of batches needs no host round trip per batch.        */
int frad_p0_overflow_scan(const double* absmax, int64_t n_frames, int32_t bits, int32_t* flag, void* stream);

/* frad_p0_digital == fourier.profile0.digital (profile0.py:46-69): unpack, NaN/Inf -> 0, inverse
 * DCT in float64, [N, C] interleaved float64 out (frame i at pcm_out + i*N*C).                    */
int frad_p0_digital(const void* payload, int64_t payload_stride, int64_t n_frames, int32_t N, int32_t C,
                    int32_t bits, uint32_t flags, double* pcm_out, void* stream);

/* A batch of equally cut CLIPS consumed and produced in place (BASELINE config 3: 4096 x 1 s clips; the reference cuts
 * every clip into frames on its own, src/libfrad/encoder.py:72-93, and its decoder returns them clip by clip):
 * frad_p0_analogue_clips == frad_p0_analogue_checked over frames (clip c, i), i < frames_per_clip, read at
 * pcm + (c*clip_stride + i*N) sample-frames -- the resident layout [n_clips, clip_stride, C] -- with the payloads dense,
 * frame c*frames_per_clip + i at payload + (c*frames_per_clip + i)*payload_stride (absmax likewise; overflow_flag may be
 * NULL).  A clip's shorter last frame is one more call with frames_per_clip = 1, its own N, and `pcm` advanced to the first
 * clip's tail.  frad_p0_digital_clips == frad_p0_digital writing frame (c, i) at pcm_out + (c*out_clip_stride + i*N)*C.
 * The N = 2048 wave kernels and the any-N Bluestein kernels address clips themselves; every other geometry goes through
 * one strided device copy to / from stream-ordered scratch.  clip_stride >= frames_per_clip*N.                       */
int frad_p0_analogue_clips(const void* pcm, int32_t pcm_dtype, int64_t n_clips, int64_t clip_stride, int32_t frames_per_clip,
                           int32_t N, int32_t C, int32_t bits, uint32_t flags, void* payload, int64_t payload_stride,
                           double* absmax, int32_t* overflow_flag, void* stream);
int frad_p0_digital_clips(const void* payload, int64_t payload_stride, int64_t n_clips, int32_t frames_per_clip, int32_t N,
                          int32_t C, int32_t bits, uint32_t flags, double* pcm_out, int64_t out_clip_stride, void* stream);

/* ---- profile 4: PCM archiving (same pack/unpack, no transform; profile4.py:14-41, 43-63) ------
 * float32/float16 PCM is cast from single precision, everything else from float64, as numpy's
 * astype does on the reference's arrays.                                                          */
int frad_p4_analogue(const void* pcm, int32_t pcm_dtype, int64_t n_frames, int32_t N, int32_t C,
                     int64_t frame_stride, int32_t bits, uint32_t flags,
                     void* payload, int64_t payload_stride, double* absmax, void* stream);
int frad_p4_digital(const void* payload, int64_t payload_stride, int64_t n_frames, int32_t N, int32_t C,
                    int32_t bits, uint32_t flags, double* pcm_out, void* stream);

/* ---- profile 1: psychoacoustic quantiser, pre-/post-entropy halves ----------------------------
 * frad_p1_analogue == fourier.profile1.analogue up to the two integer arrays (profile1.py:15-40 with
 * p1tools.py:15-44).  `N` must be a legal compact frame size (fourier/profiles.py:14-23); frame i
 * reads `n_valid` (<= N) sample-frames starting at pcm + i*frame_stride (frame_stride = hop when the
 * encoder overlaps, encoder.py:35-51) and is zero-padded to N (profile1.py:19).
 *   q   int32 [n_frames, N, C]  bin-major / channel-minor   (freqs_flat, profile1.py:34-36)
 *   tq  int32 [n_frames, 27, C] band-major / channel-minor  (thres_flat, profile1.py:38-40)
 * The Exp-Golomb-Rice stage that follows (profile1.py:43-45) is frad_p1_golomb_encode below; zlib's deflate
 * (profile1.py:50) stays on the host.                                                                  */
int frad_p1_analogue(const void* pcm, int32_t pcm_dtype, int64_t n_frames, int32_t N, int32_t C,
                     int64_t frame_stride, int32_t n_valid, int32_t bits, int32_t srate, double loss_level,
                     uint32_t flags, int32_t* q, int32_t* tq, void* stream);

/* frad_p1_digital == fourier.profile1.digital from the decoded integers on (profile1.py:65-77):
 * dequantise, spread the 27 thresholds over the bins, inverse DCT -> float64 [n_frames, N, C].
 *
 * frad_p1_overlap_add == Decoder.overlap over a batch of decoded frames (decoder.py:28-46 with the Hann ramp of
 * backend/__init__.py:3), for overlap_ratio > 1: cut = N*(ratio-1)/ratio; frame i contributes rows [0, cut) to
 * ola_out [n_frames, cut, C], its first N - cut rows cross-faded against frame i-1's tail (`prev_tail`
 * [N - cut, C] for i = 0, NULL = no previous frame: no fade); the last frame's tail is returned in `next_tail`
 * [N - cut, C] for the next batch (or the flush).                                                       */
int frad_p1_digital(const int32_t* q, const int32_t* tq, int64_t n_frames, int32_t N, int32_t C,
                    int32_t bits, int32_t srate, double* pcm_out, void* stream);
int frad_p1_overlap_add(const double* frames, int64_t n_frames, int32_t N, int32_t C, int32_t overlap_ratio,
                        const double* prev_tail, double* ola_out, double* next_tail, void* stream);

/* ---- profile 1: Exp-Golomb-Rice stage (row 8f #2) ------------------------------------------------
 * frad_p1_golomb_encode == the two exp_golomb_rice_encode calls and the struct.pack of profile1.py:43-45
 * (tools/p1tools.py:46-60): per frame the pre-deflate body  '>I' len(thres_gol) + thres_gol + freqs_gol  is
 * written at bodies + i*body_stride and its length to body_bytes[i].  body_stride >= frad_p1_golomb_bound(N, C)
 * (the worst case: k is taken from the frame's maximum, so a value costs at most k + 3 <= 35 bits), a multiple
 * of 4; `bodies` 4-byte aligned.  q / tq as frad_p1_analogue writes them.  zlib's deflate stays on the host.
 *
 * frad_rows_compact: offsets[0] = 0, offsets[i+1] = offsets[i] + row_bytes[i] (n_rows + 1 entries), and, when `out`
 * is not NULL, row i's first row_bytes[i] bytes copied to out + offsets[i] -- the batch then goes to the host as ONE
 * copy of exactly the bytes deflate needs.  The gather reads a row as aligned 32-bit words: every row needs
 * row_bytes[i] + 4 <= row_stride (frad_p1_golomb_bound includes that slack); rows that violate it are refused
 * with FRAD_E_INVALID only when row_stride < 4 -- the per-row lengths live on the device.
 *
 * frad_p1_golomb_decode == the two exp_golomb_rice_decode calls of profile1.py:59-64 (p1tools.py:62-74) plus untrim
 * (profile1.py:12-13): frame i's inflated body is bodies[offsets[i] .. offsets[i+1]); q [n_frames, N, C] and
 * tq [n_frames, 27, C] receive the decoded integers, zero-filled where the stream ends early and cut at N*C / 27*C
 * values; values outside int32 (only a corrupt stream has them) saturate.  status[i] (may be NULL) = 1 when the body
 * is shorter than its length word.  `bodies` needs 8 readable bytes after offsets[n_frames] (the streams are read as
 * aligned 32-bit words) and must not be NULL even when every body is empty.
 * Deviation on damaged streams only: missing band codes are zero-filled as INTEGERS (threshold (e/2)^0 = 1), where
 * the reference pads the dequantised thresholds with 0.0 (profile1.py:63-65) and thereby silences the affected bins. */
size_t frad_p1_golomb_bound(int32_t N, int32_t C);
int frad_p1_golomb_encode(const int32_t* q, const int32_t* tq, int64_t n_frames, int32_t N, int32_t C,
                          void* bodies, int64_t body_stride, int64_t* body_bytes, void* stream);
int frad_rows_compact(const void* rows, int64_t row_stride, const int64_t* row_bytes, int64_t n_rows, void* out,
                      int64_t* offsets, void* stream);
int frad_p1_golomb_decode(const void* bodies, const int64_t* offsets, int64_t n_frames, int32_t N, int32_t C,
                          int32_t* q, int32_t* tq, int32_t* status, void* stream);

/* ---- frame header checksum (row 8f #1) --------------------------------------------------------
 * crc_out[i] = zlib.crc32 of the `nbytes` payload bytes of frame i (at data + i*stride), the value
 * ASFH.write puts into a lossless frame's header (src/libfrad/tools/asfh.py:51-73), so a batch's
 * stream can be assembled without a host pass over the payload.                                   */
int frad_crc32_frames(const void* data, int64_t stride, int64_t n_frames, int64_t nbytes, uint32_t* crc_out, void* stream);

/* ---- decoder output conversion (R1 epilogue) and native frame-header scan (row 8f #1) ------------------------------
 * frad_from_f64 == backend.pcmformat.from_f64 followed by .astype(fmt) as the reference's caller applies it to every
 * decoded block (backend/pcmformat.py:49-62, src/decoder.py:23): float64 [n_values] -> `out_dtype` (FRAD_PCM_*), floats
 * rounded to nearest even, integers scaled and truncated; samples outside the integer range come out as numpy's astype
 * leaves them on x86-64 (cvttsd2si: low bits of the 32/64-bit conversion, 0x80..0 on overflow and for NaN).
 * flags: FRAD_RAW_BE_INTS reproduces the reference's quirk that big-endian integer formats are not recognised by
 * from_f64 (pcmformat.py:52-60 compares against native dtypes), so the unscaled float64 samples are truncated;
 * FRAD_LITTLE_ENDIAN is the payload endianness as in frad_p{0,4}_digital.
 * frad_p{0,4,1}_digital_pcm == frad_p{0,4,1}_digital with that conversion applied on the way out (profile 4: fused into
 * the unpack; profile 0 / 1: a second pass over stream-ordered scratch), pcm_out [n_frames, N, C] of out_dtype.    */
int frad_from_f64(const double* pcm, int64_t n_values, int32_t out_dtype, uint32_t flags, void* out, void* stream);
int frad_p0_digital_pcm(const void* payload, int64_t payload_stride, int64_t n_frames, int32_t N, int32_t C, int32_t bits,
                        uint32_t flags, int32_t out_dtype, void* pcm_out, void* stream);
int frad_p4_digital_pcm(const void* payload, int64_t payload_stride, int64_t n_frames, int32_t N, int32_t C, int32_t bits,
                        uint32_t flags, int32_t out_dtype, void* pcm_out, void* stream);
/* frad_p1_overlap_add_pcm == frad_p1_overlap_add with that conversion applied on the way out: ola_out [n_frames, cut, C] of
 * out_dtype in one pass over the decoded frames (next_tail stays float64: it is the next batch's input).            */
int frad_p1_overlap_add_pcm(const double* frames, int64_t n_frames, int32_t N, int32_t C, int32_t overlap_ratio, const double* prev_tail,
                            int32_t out_dtype, uint32_t flags, void* ola_out, double* next_tail, void* stream);
int frad_p1_digital_pcm(const int32_t* q, const int32_t* tq, int64_t n_frames, int32_t N, int32_t C, int32_t bits,
                        int32_t srate, int32_t out_dtype, uint32_t flags, void* pcm_out, void* stream);

/* frad_asfh_scan: HOST code (no device involved).  One pass over a FrAD byte stream from `start`: resynchronises on
 * FRM_SIGN and parses every header as ASFH.read does (tools/asfh.py:98-134; the search as decoder.py:82-90), filling
 * frames[0 .. return value).  Stops at the first header or payload that is not completely inside the buffer, at
 * max_frames, or at the end; *next_pos = where the next call (with more data appended) must resume, *stop_reason says
 * why.  Force-flush headers are table rows with force_flush = 1 and no payload.  A compact header whose sample-rate
 * index is not in the table (>= 12; the reference raises) is reported with srate = 0.  Returns the row count or
 * FRAD_E_INVALID. */
typedef struct frad_frame_info {
    int64_t header_off, payload_off, payload_bytes;
    int32_t profile, ecc, little_endian, depth_idx, channels, srate, fsize, overlap_ratio, ecc_dsize, ecc_codesize, force_flush;
    uint32_t crc;                                 /* crc32 of the payload (lossless) / crc16 (compact with ECC) as stored */
} frad_frame_info;
enum { FRAD_SCAN_END = 0, FRAD_SCAN_PARTIAL_HEADER = 1, FRAD_SCAN_PARTIAL_PAYLOAD = 2, FRAD_SCAN_TABLE_FULL = 3 };
int64_t frad_asfh_scan(const void* stream_bytes, int64_t nbytes, int64_t start, frad_frame_info* frames, int64_t max_frames,
                       int64_t* next_pos, int32_t* stop_reason);

/* ---- measurement aid (not on the codec path) ----------------------------------------------------
 * dst[0, nbytes) = src[0, nbytes): device-to-device, 16 bytes per lane, the access shape of the
 * transform kernels.  bench.py times it as the "achievable HBM bandwidth" yardstick next to the
 * codec kernels.  nbytes and both pointers must be multiples of 16.                                 */
int frad_bench_copy(const void* src, void* dst, int64_t nbytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FRAD_HIP_H */
