// isf_decode.hip -- the small gather / scatter kernels around the two top-k selections (instance mining, A12-A13;
// detection head, 8f #1) and SURVEY 8f #1, second half: TransFusionHeadV2.get_bboxes with nms_type=None (the shipped
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

// ---------------------------------------------------------------------------------------- proposal initialisation
// transfusion_head_v2.py:806-842 after the top-k: the reference (and rounds 2-5 here) spend ~15 small torch ops on
// labels = top // HW, query_pos = bev_pos.gather(top % HW), query = feat.gather + class_encoding(one_hot(labels)),
// the first layer's query position embedding, query + pos, and query_heatmap_score = heatmap.gather.  One launch:
// block = one proposal, threads = channels.
__global__ __launch_bounds__(128) void head_query_init_kernel(
    const int32_t* __restrict__ top_index, const int32_t* __restrict__ top_raw, int P, int HW, int E, int C,
    const float* __restrict__ feat_tok, const int64_t* __restrict__ tok_of_cell, const float* __restrict__ class_table,
    const float* __restrict__ qpe_table, const float* __restrict__ bev_pos, const float* __restrict__ masked,
    float* __restrict__ query, float* __restrict__ qpe, float* __restrict__ x, float* __restrict__ query_pos,
    int64_t* __restrict__ top_index64, int64_t* __restrict__ labels, float* __restrict__ query_score) {
  const int bp = blockIdx.x, b = bp / P, p = bp - b * P;
  const int cell = top_index[bp];
  const int cls = top_raw[bp] / HW;
  const long long row = (long long)b * HW + (tok_of_cell ? tok_of_cell[cell] : (long long)cell);
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const float q = feat_tok[row * E + e] + class_table[(size_t)cls * E + e];
    query[(size_t)bp * E + e] = q;
    if (qpe_table) {
      const float pe = qpe_table[(size_t)cell * E + e];
      qpe[(size_t)bp * E + e] = pe;
      x[(size_t)bp * E + e] = q + pe;
    }
  }
  if (threadIdx.x < 2) query_pos[(size_t)bp * 2 + threadIdx.x] = bev_pos[(size_t)cell * 2 + threadIdx.x];
  if (threadIdx.x == 2) top_index64[bp] = cell;
  if (threadIdx.x == 3) labels[bp] = cls;
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    query_score[((size_t)b * C + c) * P + p] = masked[((size_t)b * C + c) * HW + cell];
}

// ---------------------------------------------------------------------------------------- prediction outputs
// The FFN heads' last GEMMs leave [B*P, columns] token-major blocks; the reference's result dict holds one [B, c, P]
// tensor per output, `center` offset by the query positions (:883) and handed on as the next layer's positions (:885).
// (Six transposes + an add + a clone per decoder layer before.)
static constexpr int kMaxHeads = 8;
struct HeadScatter {
  const float* src[kMaxHeads];
  float* dst[kMaxHeads];
  int ld[kMaxHeads], col0[kMaxHeads], ch[kMaxHeads], first[kMaxHeads + 1];   // first[h] = rows of dst before head h
  int n, center;
};

__global__ __launch_bounds__(256) void head_scatter_kernel(HeadScatter hs, int B, int P, const float* __restrict__ query_pos,
                                                           float* __restrict__ query_pos_next) {
  const int total = hs.first[hs.n] * B * P;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int p = i % P, r = (i / P) % hs.first[hs.n], b = i / (P * hs.first[hs.n]);
  int h = 0;
#pragma unroll
  for (int k = 1; k < kMaxHeads; ++k) h += (k < hs.n && r >= hs.first[k]) ? 1 : 0;
  const int c = r - hs.first[h];
  float v = hs.src[h][((size_t)b * P + p) * hs.ld[h] + hs.col0[h] + c];
  if (h == hs.center) {
    v += query_pos[((size_t)b * P + p) * 2 + c];
    if (query_pos_next) query_pos_next[((size_t)b * P + p) * 2 + c] = v;
  }
  hs.dst[h][((size_t)b * hs.ch[h] + c) * P + p] = v;
}

// ---------------------------------------------------------------------------------------- mined instances
// fusion_encoder.py:1133-1141 + InsContextAtt.forward's preamble (:800-812): the instances' feature columns, their
// create_2D_grid positions (raw and normalised), their position embedding (a row of the per-cell table) and features +
// embedding -- ~27 small torch ops per forward before.  top [B, Q] = cell y'*S + x' of the TRANSPOSED map (what
// isf_instance_topk returns for the transposed heat-map); scene [B, E, S, S] holds the un-transposed orientation, where
// that cell is x'*S + y'.  Block = one instance, threads = channels.
__global__ __launch_bounds__(128) void instance_gather_kernel(
    const int32_t* __restrict__ top, int Q, int S, int E, const float* __restrict__ scene,
    const float* __restrict__ qpe_table, int64_t* __restrict__ top64, int64_t* __restrict__ cell64,
    float* __restrict__ tokens, float* __restrict__ qpe, float* __restrict__ tokens_pos, float* __restrict__ query_pos,
    float* __restrict__ ref) {
  const int bq = blockIdx.x, b = bq / Q;
  const int t = top[bq];
  const int xq = t % S, yq = t / S;   // (x', y')
  const int cell = xq * S + yq;
  const size_t HW = (size_t)S * S;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const float v = scene[((size_t)b * E + e) * HW + cell];
    const float pe = qpe_table[(size_t)cell * E + e];
    tokens[(size_t)bq * E + e] = v;
    qpe[(size_t)bq * E + e] = pe;
    tokens_pos[(size_t)bq * E + e] = v + pe;
  }
  if (threadIdx.x == 0) {
    const float px = (float)xq + 0.5f, py = (float)yq + 0.5f;
    query_pos[(size_t)bq * 2] = px;
    query_pos[(size_t)bq * 2 + 1] = py;
    ref[(size_t)bq * 2] = px / (float)S;
    ref[(size_t)bq * 2 + 1] = py / (float)S;
    top64[bq] = t;
    cell64[bq] = cell;
  }
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

int isf_head_query_init(const int32_t* top_index, const int32_t* top_raw, int batch_size, int num_proposals, int hw,
                        int embed, int num_classes, const float* feat_tok, const int64_t* tok_of_cell,
                        const float* class_table, const float* qpe_table, const float* bev_pos, const float* masked,
                        float* query, float* qpe, float* x, float* query_pos, int64_t* top_index64,
                        int64_t* query_labels, float* query_score, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && num_proposals >= 0 && hw > 0 && embed > 0 && num_classes > 0, ISF_ERR_ARG,
              "head_query_init: bad sizes (B %d, proposals %d, HW %d, E %d, classes %d)", batch_size, num_proposals, hw,
              embed, num_classes);
  if (batch_size * num_proposals == 0) return ISF_OK;
  ISF_REQUIRE(top_index && top_raw && feat_tok && class_table && bev_pos && masked && query && query_pos && top_index64 &&
                  query_labels && query_score && (!qpe_table || (qpe && x)), ISF_ERR_ARG, "head_query_init: null pointer");
  hipLaunchKernelGGL(head_query_init_kernel, dim3(batch_size * num_proposals), dim3(128), 0, as_stream(stream), top_index,
                     top_raw, num_proposals, hw, embed, num_classes, feat_tok, tok_of_cell, class_table, qpe_table, bev_pos,
                     masked, query, qpe, x, query_pos, top_index64, query_labels, query_score);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_head_scatter_predictions(int num_heads, const float* const* src, const int* src_ld, const int* col0,
                                 const int* channels, float* const* dst, int center_head, const float* query_pos,
                                 float* query_pos_next, int batch_size, int num_proposals, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_heads > 0 && num_heads <= kMaxHeads && batch_size >= 0 && num_proposals >= 0 && src && src_ld && col0 &&
                  channels && dst, ISF_ERR_ARG, "head_scatter_predictions: bad arguments (%d heads)", num_heads);
  ISF_REQUIRE(center_head < num_heads && (center_head < 0 || (query_pos && channels[center_head] == 2)), ISF_ERR_ARG,
              "head_scatter_predictions: the centre output needs query_pos and two channels");
  if (batch_size * num_proposals == 0) return ISF_OK;
  HeadScatter hs;
  hs.n = num_heads;
  hs.center = center_head;
  hs.first[0] = 0;
  for (int h = 0; h < kMaxHeads; ++h) {
    const bool on = h < num_heads;
    hs.src[h] = on ? src[h] : nullptr;
    hs.dst[h] = on ? dst[h] : nullptr;
    hs.ld[h] = on ? src_ld[h] : 0;
    hs.col0[h] = on ? col0[h] : 0;
    hs.ch[h] = on ? channels[h] : 0;
    ISF_REQUIRE(!on || (src[h] && dst[h] && channels[h] > 0 && col0[h] >= 0 && src_ld[h] >= col0[h] + channels[h]),
                ISF_ERR_ARG, "head_scatter_predictions: head %d: columns [%d, %d) of %d", h, on ? col0[h] : 0,
                on ? col0[h] + channels[h] : 0, on ? src_ld[h] : 0);
    hs.first[h + 1] = hs.first[h] + hs.ch[h];
  }
  for (int h = num_heads; h < kMaxHeads; ++h) hs.first[h + 1] = hs.first[num_heads];
  const int total = hs.first[num_heads] * batch_size * num_proposals;
  hipLaunchKernelGGL(head_scatter_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), hs, batch_size,
                     num_proposals, query_pos, query_pos_next);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_instance_gather(const int32_t* top, int batch_size, int num_instances, int bev_size, int embed,
                        const float* scene, const float* qpe_table, int64_t* top64, int64_t* cell64, float* tokens,
                        float* qpe, float* tokens_pos, float* query_pos, float* ref, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && num_instances >= 0 && bev_size > 0 && embed > 0, ISF_ERR_ARG,
              "instance_gather: bad sizes (B %d, instances %d, S %d, E %d)", batch_size, num_instances, bev_size, embed);
  if (batch_size * num_instances == 0) return ISF_OK;
  ISF_REQUIRE(top && scene && qpe_table && top64 && cell64 && tokens && qpe && tokens_pos && query_pos && ref,
              ISF_ERR_ARG, "instance_gather: null pointer");
  hipLaunchKernelGGL(instance_gather_kernel, dim3(batch_size * num_instances), dim3(128), 0, as_stream(stream), top,
                     num_instances, bev_size, embed, scene, qpe_table, top64, cell64, tokens, qpe, tokens_pos, query_pos,
                     ref);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
