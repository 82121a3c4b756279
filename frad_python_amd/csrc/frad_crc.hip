// frad_crc.hip -- CRC-32 of every frame payload of a batch, on the device (row 8f #1: the frame header's
// checksum, tools/asfh.py:69-76 == zlib.crc32 of the payload in the reference's ASFH.write).
//
// One wave per frame.  The payload is staged 16 KiB at a time through LDS with coalesced 16-byte loads;
// every lane then runs the byte-table CRC over its own 256-byte chunk (chunks padded to 260 bytes in LDS:
// lane stride 65 words, no bank conflicts), multiplies its register by x^(8 * bytes after the chunk) mod P
// -- a product of precomputed 32 x 32 GF(2) matrices for the powers of two, i.e. zlib's crc32_combine --
// and the lanes' results are XOR-ed.  Reflected polynomial 0xEDB88320, initial value and final XOR ~0.
#include "../../include/frad_hip.h"
#include "frad_launch.hpp"

#include <map>
#include <mutex>
#include <vector>

namespace frad {

constexpr int CRC_CHUNK = 256, CRC_PAD = CRC_CHUNK + 4, CRC_ROW = 64 * CRC_CHUNK;   // bytes per lane / per staging round
constexpr int CRC_TAB_WORDS = 256 + 32 * 32;                                       // byte table + shift matrices

template <int DUMMY>
__global__ void __launch_bounds__(256) k_crc32_frames(const unsigned char* __restrict__ data, long long stride, long long n_frames,
                                                      long long nbytes, const uint32_t* __restrict__ tables,
                                                      uint32_t* __restrict__ out, int aligned) {
    FRAD_DYN_SMEM(smem);
    uint32_t* tab = reinterpret_cast<uint32_t*>(smem);                 // [256] byte table, then [32][32] matrices
    for (int i = threadIdx.x; i < CRC_TAB_WORDS; i += blockDim.x) tab[i] = tables[i];
    __syncthreads();
    const uint32_t* mats = tab + 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    unsigned char* stage = smem + CRC_TAB_WORDS * 4 + wave * (64 * CRC_PAD);
    for (long long f = (long long)blockIdx.x * wpb + wave; f < n_frames; f += (long long)gridDim.x * wpb) {
        const unsigned char* src = data + f * stride;
        uint32_t acc = nbytes == 0 ? 0xFFFFFFFFu : 0u;          // crc32(b"") == 0
        for (long long base = 0; base < nbytes; base += CRC_ROW) {
            const long long left = nbytes - base;
            const int row = left < CRC_ROW ? (int)left : CRC_ROW;       // bytes staged this round
            // coalesced stage-in: 16-byte pieces, lane-contiguous in global memory, chunk-padded in LDS
            for (int p = lane; p * 16 < row; p += 64) {
                const int off = p * 16, ch = off / CRC_CHUNK, in = off - ch * CRC_CHUNK;
                unsigned char* dst = stage + ch * CRC_PAD + in;
                if (aligned && off + 16 <= row) {
                    uint32_t w[4];
                    load_words<4>(src + base + off, w);
#pragma unroll
                    for (int i = 0; i < 4; ++i) reinterpret_cast<uint32_t*>(dst)[i] = w[i];
                } else {
                    for (int i = 0; i < 16 && off + i < row; ++i) dst[i] = src[base + off + i];
                }
            }
            team_sync<64>();
            const int c0 = lane * CRC_CHUNK;
            if (c0 < row) {
                const int len = row - c0 < CRC_CHUNK ? row - c0 : CRC_CHUNK;
                uint32_t s = (base == 0 && lane == 0) ? 0xFFFFFFFFu : 0u;
                const unsigned char* ch = stage + lane * CRC_PAD;
                int i = 0;
                for (; i + 4 <= len; i += 4) {
                    uint32_t w = *reinterpret_cast<const uint32_t*>(ch + i);
#pragma unroll
                    for (int b = 0; b < 4; ++b) { s = tab[(s ^ w) & 0xffu] ^ (s >> 8); w >>= 8; }
                }
                for (; i < len; ++i) s = tab[(s ^ ch[i]) & 0xffu] ^ (s >> 8);
                // advance over the bytes that follow this chunk: s * x^(8 R) mod P
                unsigned long long R = (unsigned long long)(nbytes - (base + c0 + len));
                for (int k = 0; R != 0; ++k, R >>= 1) {
                    if (R & 1ull) {
                        const uint32_t* m = mats + k * 32;
                        uint32_t r = 0;
#pragma unroll 8
                        for (int j = 0; j < 32; ++j) r ^= m[j] & (0u - ((s >> j) & 1u));
                        s = r;
                    }
                }
                acc ^= s;
            }
            team_sync<64>();
        }
        if (nbytes != 0) acc = (uint32_t)wave_allreduce_u64((u64)acc, [](u64 a, u64 b) { return a ^ b; });
        if (lane == 0) out[f] = acc ^ 0xFFFFFFFFu;
    }
}

namespace {

std::mutex g_mu;
std::map<int, uint32_t*> g_tab;                              // device -> tables
thread_local int g_last = 0;
#define CCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_last = (int)e_; return FRAD_E_HIP; } } while (0)

int get_crc_tables(const uint32_t** out) {
    int dev = 0; CCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_tab.find(dev);
    if (it != g_tab.end()) { *out = it->second; return FRAD_OK; }
    std::vector<uint32_t> h(CRC_TAB_WORDS);
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        h[n] = c;
    }
    uint32_t* m = h.data() + 256;
    for (int j = 0; j < 32; ++j) { const uint32_t x = 1u << j; m[j] = h[x & 0xffu] ^ (x >> 8); }          // one zero byte
    for (int k = 1; k < 32; ++k)                             // squaring: 2^k bytes = (2^(k-1) bytes) twice
        for (int j = 0; j < 32; ++j) {
            const uint32_t x = m[(k - 1) * 32 + j];
            uint32_t r = 0;
            for (int b = 0; b < 32; ++b) if ((x >> b) & 1u) r ^= m[(k - 1) * 32 + b];
            m[k * 32 + j] = r;
        }
    uint32_t* d = nullptr;
    CCHK(hipMalloc(&d, h.size() * sizeof(uint32_t)));
    CCHK(hipMemcpy(d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    g_tab[dev] = d; *out = d;
    return FRAD_OK;
}

}  // namespace

int crc_last_hip_error() { return g_last; }
void crc_clear() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_tab) (void)hipFree(kv.second);
    g_tab.clear();
}

}  // namespace frad

extern "C" int frad_crc32_frames(const void* data, int64_t stride, int64_t n_frames, int64_t nbytes, uint32_t* crc_out, void* stream) {
    using namespace frad;
    if (n_frames < 0 || nbytes < 0 || stride < 0) return FRAD_E_INVALID;
    if (n_frames == 0) return FRAD_OK;
    if (!crc_out || (nbytes > 0 && !data)) return FRAD_E_INVALID;
    const uint32_t* tables = nullptr;
    const int rc = get_crc_tables(&tables);
    if (rc != FRAD_OK) return rc;
    const int aligned = ((reinterpret_cast<uintptr_t>(data) | (uintptr_t)stride) & 15u) == 0 ? 1 : 0;
    const size_t lds = (size_t)CRC_TAB_WORDS * 4 + 4 * 64 * (size_t)CRC_PAD;
    long long blocks = (n_frames + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    allow_lds(k_crc32_frames<0>, lds);
    hipLaunchKernelGGL(k_crc32_frames<0>, dim3((unsigned)blocks), dim3(256), lds, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned char*>(data), (long long)stride, (long long)n_frames, (long long)nbytes, tables, crc_out, aligned);
    if (hipGetLastError() != hipSuccess) return FRAD_E_HIP;
    return FRAD_OK;
}
