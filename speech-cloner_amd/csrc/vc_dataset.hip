// Row gather for the on-device feature cache (sound_ds.py).
//
// The reference keeps per-utterance feature matrices in an h5py file and cuts training windows
// out of it on the host (/root/reference/sound_ds.py:262-350, ARCTIC_reader.py:277-362,
// TIMIT_reader.py:474-523).  Here the features of a whole corpus stay in HBM as one ragged arena
// [total_frames, C]; packing front-end output into the arena and cutting a batch of windows are
// the same operation: dst[r, :] = src[index[r], :], or a constant pad row where index[r] < 0.
#include "vc_common.h"

namespace {

__global__ void __launch_bounds__(256)
gather_rows_kernel(const uint32_t* __restrict__ src, const int64_t* __restrict__ index,
                   const uint32_t* __restrict__ pad_row, int64_t total, int words, uint32_t* __restrict__ dst) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / words;
        const int w = (int)(e - r * words);
        const int64_t s = index[r];
        uint32_t v = pad_row ? pad_row[w] : 0u;
        if (s >= 0) v = src[s * words + w];
        dst[e] = v;
    }
}

}  // namespace

extern "C" int vc_gather_rows(const void* d_src, const int64_t* d_index, const void* d_pad_row, int64_t n_rows,
                              int32_t row_bytes, void* d_dst, void* stream) {
    VC_REQUIRE(d_src && d_index && d_dst, "vc_gather_rows: NULL argument");
    VC_REQUIRE(n_rows >= 0 && row_bytes > 0 && (row_bytes % 4) == 0, "vc_gather_rows: row_bytes must be a positive multiple of 4");
    if (n_rows == 0) return VC_OK;
    const int words = row_bytes / 4;
    const int64_t total = n_rows * words;
    const int64_t want = (total + 255) / 256;
    const unsigned grid = (unsigned)(want < 65536 ? want : 65536);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint32_t*>(d_src), d_index, reinterpret_cast<const uint32_t*>(d_pad_row),
                       total, words, reinterpret_cast<uint32_t*>(d_dst));
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}
