// Pieces shared by the two K_sinc translation units (sinc.hip: block kernels; sinc2.hip: the streaming kernel).
#pragma once
#include "par_common.h"
#include "pos_plan.h"

namespace par {

// sin(pi*x), cos(pi*x) on [-0.5, 0.5]; Taylor in (pi*x), abs error < 1e-7 at the interval ends.
__device__ __forceinline__ float sinpi_half(float x) {
  const float z = x * x;
  float p = -0.00737043094f;               // -pi^11/11!
  p = fmaf(p, z, 0.0821458866f);           //  pi^9/9!
  p = fmaf(p, z, -0.599264529f);           // -pi^7/7!
  p = fmaf(p, z, 2.55016404f);             //  pi^5/5!
  p = fmaf(p, z, -5.16771278f);            // -pi^3/3!
  p = fmaf(p, z, 3.14159265f);             //  pi
  return p * x;
}
__device__ __forceinline__ float cospi_half(float x) {
  const float z = x * x;
  float p = 0.00192957431f;                //  pi^12/12!
  p = fmaf(p, z, -0.0258068914f);          // -pi^10/10!
  p = fmaf(p, z, 0.235330630f);            //  pi^8/8!
  p = fmaf(p, z, -1.33526277f);            // -pi^6/6!
  p = fmaf(p, z, 4.05871213f);             //  pi^4/4!
  p = fmaf(p, z, -4.93480220f);            // -pi^2/2!
  p = fmaf(p, z, 1.0f);
  return p;
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// What the fused kernels read of a plan (views into the caller's work / aux buffers, pos_plan.h).
struct FusedArgs {
  const double* speeds;
  const int64_t* seg_start;
  const double* seg_off;
  const double* ck;
  const int64_t* tile_seg;
  const SegFast* seg_fast;
  const TileHdr* hdr;
  const BlockRec* rec;
  const BlockRec2* rec2;
  int64_t nseg;
  int* redo_count;           // streaming kernel -> block kernel: tiles to do the old way ([0] = how many)
  int* redo_list;
};

// Streaming kernel (sinc2.hip): NT = 32; nch = 1: a mono signal on unit strides, nch = 2: an interleaved stereo file (sig / out
// point at the left channel's first sample; len_in / len_out count frames); nch = 2 with pick_out_stride > 0: ONE channel of a
// two-channel interleaved file (sig = that channel's first sample, input stride 2; outputs pick_out_stride elements apart).
// Tiles it does not take are appended to fa.redo_list.
struct TapModes;
int launch_sinc_stream(int device, int64_t len_out, const float* sig, int64_t len_in, float* out, const FusedArgs& fa,
                       const float4* tab, const TapModes& tmd, hipStream_t s, int nch, int64_t pick_out_stride = 0);

// ... and several files of one form in one launch per kernel kind (k_sinc_pipe_n; at most kFusedBatchMax)
constexpr int kFusedBatchMax = 8;
struct StreamItem {
  int64_t len_out;
  const float* sig;
  int64_t len_in;
  float* out;
  FusedArgs fa;
};
int launch_sinc_stream_batch(int device, int n, const StreamItem* items, const float4* tab, const TapModes& tmd, hipStream_t s, int nch);

}  // namespace par
