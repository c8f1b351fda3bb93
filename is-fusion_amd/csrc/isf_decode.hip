// isf_decode.hip -- SURVEY 8f #1, second half: TransFusionHeadV2.get_bboxes with nms_type=None (the shipped
// nuScenes test_cfg) = proposal scoring (transfusion_head_v2.py:1288-1294) + TransFusionBBoxCoder.decode
// (core/bbox/coders/transfusion_bbox_coder.py:39-124) + the centre-range / score filter (:100-118).
//
// The reference runs ~25 small torch ops per frame over [B, *, 200] tensors plus a boolean-mask gather per sample
// (a device->host sync each).  Here: one wave per sample, one launch, no sync; lanes own proposals, the kept ones
// are compacted in proposal order with ballot + popcount (the reference's mask indexing keeps that order).
#include "isf_common.h"

namespace isf {

__device__ __forceinline__ float decode_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

struct DecodeParams {
  float cell_x, cell_y;   // out_size_factor * voxel_size
  float org_x, org_y;     // pc_range[:2]
  float lo[3], hi[3];     // post_center_range
  float score_threshold;
  int use_threshold;      // `if self.score_threshold:` (:108) -- a threshold of 0.0 is NOT applied
};

__global__ __launch_bounds__(64) void decode_boxes_kernel(
    const float* __restrict__ heatmap, const float* __restrict__ query_score, const int64_t* __restrict__ labels,
    const float* __restrict__ center, const float* __restrict__ height, const float* __restrict__ dim,
    const float* __restrict__ rot, const float* __restrict__ vel, int C, int P, int ld, DecodeParams prm,
    float* __restrict__ boxes, float* __restrict__ scores, int32_t* __restrict__ out_labels,
    int32_t* __restrict__ counts) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int code = vel ? 9 : 7;
  const float* hm = heatmap + (size_t)b * C * ld;
  const float* qs = query_score + (size_t)b * C * ld;
  int kept = 0;
  for (int p0 = 0; p0 < P; p0 += 64) {
    const int p = p0 + lane;
    bool keep = false;
    float box[9], best = 0.f;
    int arg = 0;
    if (p < P) {
      // score = sigmoid(heatmap) * query_heatmap_score * one_hot(label); max / argmax over classes, first maximum wins
      const int lab = (int)labels[(size_t)b * P + p];
      for (int c = 0; c < C; ++c) {
        const float s = decode_sigmoid(hm[(size_t)c * ld + p]) * qs[(size_t)c * ld + p] * (c == lab ? 1.f : 0.f);
        if (c == 0 || s > best) { best = s; arg = c; }
      }
      const float* ce = center + (size_t)b * 2 * ld;
      const float* di = dim + (size_t)b * 3 * ld;
      const float* ro = rot + (size_t)b * 2 * ld;
      box[0] = ce[p] * prm.cell_x + prm.org_x;
      box[1] = ce[ld + p] * prm.cell_y + prm.org_y;
      box[3] = expf(di[p]);
      box[4] = expf(di[ld + p]);
      box[5] = expf(di[2 * ld + p]);
      box[2] = height[(size_t)b * ld + p] - box[5] * 0.5f;   // gravity centre -> bottom centre
      box[6] = atan2f(ro[p], ro[ld + p]);
      if (vel) {
        box[7] = vel[(size_t)b * 2 * ld + p];
        box[8] = vel[(size_t)b * 2 * ld + ld + p];
      }
      keep = true;
#pragma unroll
      for (int a = 0; a < 3; ++a) keep = keep && box[a] >= prm.lo[a] && box[a] <= prm.hi[a];
      if (prm.use_threshold) keep = keep && best > prm.score_threshold;
    }
    const unsigned long long mask = __ballot(keep);
    if (keep) {
      const int at = kept + __popcll(mask & ((1ull << lane) - 1ull));
      float* o = boxes + ((size_t)b * P + at) * code;
      for (int a = 0; a < code; ++a) o[a] = box[a];
      scores[(size_t)b * P + at] = best;
      out_labels[(size_t)b * P + at] = arg;
    }
    kept += __popcll(mask);
  }
  if (lane == 0) counts[b] = kept;
}

}  // namespace isf

extern "C" {

int isf_decode_boxes(const float* heatmap, const float* query_score, const int64_t* query_labels, const float* center,
                     const float* height, const float* dim, const float* rot, const float* vel, int batch_size,
                     int num_classes, int num_proposals, int ld, const float* coder, float* boxes, float* scores,
                     int32_t* labels, int32_t* counts, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && num_classes > 0 && num_proposals >= 0 && ld >= num_proposals, ISF_ERR_ARG,
              "decode_boxes: bad sizes (B %d, classes %d, proposals %d, ld %d)", batch_size, num_classes,
              num_proposals, ld);
  if (batch_size == 0) return ISF_OK;
  ISF_REQUIRE(heatmap && query_score && query_labels && center && height && dim && rot && coder && boxes && scores &&
                  labels && counts, ISF_ERR_ARG, "decode_boxes: null pointer");
  DecodeParams prm;
  prm.cell_x = coder[0];
  prm.cell_y = coder[1];
  prm.org_x = coder[2];
  prm.org_y = coder[3];
  for (int a = 0; a < 3; ++a) {
    prm.lo[a] = coder[4 + a];
    prm.hi[a] = coder[7 + a];
  }
  prm.score_threshold = coder[10];
  prm.use_threshold = coder[11] != 0.f;
  hipLaunchKernelGGL(decode_boxes_kernel, dim3(batch_size), dim3(64), 0, as_stream(stream), heatmap, query_score,
                     query_labels, center, height, dim, rot, vel, num_classes, num_proposals, ld, prm, boxes, scores,
                     labels, counts);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
