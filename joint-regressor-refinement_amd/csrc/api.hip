// C ABI (include/jrr.h): model upload, engine/workspace planning and the launch sequences.
#include <algorithm>
#include <array>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <tuple>
#include <vector>

#include "jrr_common.h"
#include "kernels.h"

using namespace jrr;

static thread_local char g_err[512] = "";
void jrr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* jrr_last_error(void) { return g_err; }
extern "C" int jrr_version(void) { return 106; }      // 100 + round: entry points were added in rounds 2-6, none changed or removed

#define CHECK_LAUNCH()                                                            \
  do {                                                                            \
    hipError_t _e = hipGetLastError();                                            \
    if (_e != hipSuccess) {                                                       \
      jrr_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return JRR_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)

// =============================================================================================
// model
// =============================================================================================
constexpr int MAX_FACES = 14336;       // the rasteriser's capacity: 1024 threads x 14 faces (sil.hip)
static size_t model_floats() {
  const size_t nDk = (size_t)VT * KFP * 96, nDn = (size_t)3 * VP * KFP, nDq = nDn, nWjv = (size_t)VT * NJ * 32, nWvj = (size_t)VT * 1024;
  const size_t nJt = 72 + 24, nJS = 720 + 16, nWc = (size_t)(VT + 1) * NJ * 32, nJl = (size_t)VT * NJ + (size_t)VT + 8, nPerm = (size_t)VP + 6912;
  const size_t nW16 = (size_t)VT * 16 * 36, nSeg = (size_t)VT * 32;
  return nDk + nDn + nDq + nWjv + nWvj + nJt + nJS + nWc + nJl + nPerm + nW16 + nSeg;
}
extern "C" size_t jrr_model_bytes(void) { return round_up(model_floats() * sizeof(float), 256) + (size_t)2 * MAX_FACES * (3 + 2) * sizeof(int32_t); }

extern "C" int jrr_model_create(const float* vt, const float* sd, const float* pd, const float* Jr, const float* W,
                                const int32_t* parents, jrr_model_t** out) {
  return jrr_model_create_in(vt, sd, pd, Jr, W, parents, nullptr, 0, out);
}

extern "C" int jrr_model_create_in(const float* vt, const float* sd, const float* pd, const float* Jr, const float* W,
                                   const int32_t* parents, void* buffer_dev, size_t buffer_bytes, jrr_model_t** out) {
  return jrr_model_create_hinted(vt, sd, pd, Jr, W, parents, nullptr, 0, buffer_dev, buffer_bytes, out);
}

extern "C" int jrr_model_create_hinted(const float* vt, const float* sd, const float* pd, const float* Jr, const float* W,
                                       const int32_t* parents, const int32_t* hint_vertices, int n_hint, void* buffer_dev,
                                       size_t buffer_bytes, jrr_model_t** out) {
  if (n_hint < 0 || n_hint > V || (n_hint > 0 && !hint_vertices)) { jrr_set_error("jrr_model_create_hinted: bad hint"); return JRR_ERR_ARG; }
  for (int i = 0; i < n_hint; ++i)
    if (hint_vertices[i] < 0 || hint_vertices[i] >= V) { jrr_set_error("jrr_model_create_hinted: hint vertex %d out of range", hint_vertices[i]); return JRR_ERR_ARG; }
  if (!vt || !sd || !pd || !Jr || !W || !parents || !out) { jrr_set_error("jrr_model_create: null argument"); return JRR_ERR_ARG; }
  if (buffer_dev && (buffer_bytes < jrr_model_bytes() || ((uintptr_t)buffer_dev & 255) != 0)) {
    jrr_set_error("jrr_model_create_in: the model buffer needs jrr_model_bytes() = %zu bytes, 256-byte aligned", jrr_model_bytes());
    return JRR_ERR_WORKSPACE;
  }
  for (int j = 0; j < NJ; ++j)
    if (parents[j] >= j || (j > 0 && parents[j] < 0)) { jrr_set_error("parents[%d]=%d is not a topologically ordered tree", j, parents[j]); return JRR_ERR_ARG; }
  const size_t nDk = (size_t)VT * KFP * 96, nDn = (size_t)3 * VP * KFP, nDq = nDn, nWjv = (size_t)VT * NJ * 32, nWvj = (size_t)VT * 1024;
  const size_t nJt = 72 + 24, nJS = 720 + 16;   // padded to keep 16-byte alignment of what follows
  const size_t nWc = (size_t)(VT + 1) * NJ * 32, nJl = (size_t)VT * NJ + (size_t)VT + 8, nPerm = (size_t)VP + 6912;   // jl [VT][24] + tnj [VT] (+ pad); p2v [VP], v2p [V] padded
  const size_t nW16 = (size_t)VT * 16 * 36, nSeg = (size_t)VT * 32;                      // segid [VT] (padded to 16 per tile), segj [VT][16]
  std::vector<float> h(nDk + nDn + nDq + nWjv + nWvj + nJt + nJS + nWc + nJl + nPerm + nW16 + nSeg, 0.f);
  float* Dk = h.data();
  float* Dn = Dk + nDk;
  float* Dq = Dn + nDn;
  float* Wjv = Dq + nDq;
  float* Wvj = Wjv + nWjv;
  float* Jt = Wvj + nWvj;
  float* JS = Jt + nJt;
  float* Wc = JS + nJS;
  int32_t* Jl = reinterpret_cast<int32_t*>(Wc + nWc);
  int32_t* Tnj = Jl + (size_t)VT * NJ;
  int32_t* P2V = Jl + nJl;
  int32_t* V2P = P2V + VP;
  float* W16 = reinterpret_cast<float*>(P2V + nPerm);
  int32_t* SegId = reinterpret_cast<int32_t*>(W16 + nW16);
  int32_t* SegJ = SegId + (size_t)VT * 16;
  // ---- internal vertex order (jrr_common.h): the file order unless it does not fit the joint-sparse kernels and the
  //      joint-sorted order does (or JRR_VERTEX_ORDER=sorted asks for it: tests) ----
  // cost of an order for the joint-sparse kernels: matrix instructions of the forward kernel with per-tile classes (a tile
  // above the slot count pays a second pass), infinite when a tile exceeds the backward kernel's 16-joint window
  auto order_cost = [&](const std::vector<int>& order) {
    long cost8 = 0, cost12 = 0;
    for (int t = 0; t < VT; ++t) {
      bool used[NJ] = {false};
      for (int vv = 0; vv < 32; ++vv) {
        const int p_ = t * 32 + vv;
        if (p_ >= V) break;
        for (int j = 0; j < NJ; ++j) used[j] = used[j] || W[(size_t)order[p_] * NJ + j] != 0.f;
      }
      int n = 0;
      for (int j = 0; j < NJ; ++j) n += used[j];
      if (n > KJS_TILE_MAX) return (long)1 << 40;
      cost8 += 423 + (n > 8 ? 78 : 0);
      cost12 += 447 + (n > 12 ? 102 : 0);
    }
    return std::min(cost8, cost12);
  };
  std::vector<int> order(V);
  for (int v = 0; v < V; ++v) order[v] = v;
  bool permuted = false;
  int hint_applied = 0;
  {
    const char* ask = getenv("JRR_VERTEX_ORDER");
    const bool force = ask && strcmp(ask, "sorted") == 0;
    const long cost_file = order_cost(order);
    if (force || n_hint > 0 || cost_file > (long)VT * 423) {      // some tile of the file order is wide: would a joint-sorted order be cheaper?
      // joints of each vertex by descending weight (dominant joint first)
      std::vector<std::array<int, 4>> inf4(V);
      for (int v = 0; v < V; ++v) {
        std::vector<std::pair<float, int>> inf;
        for (int j = 0; j < NJ; ++j) if (W[(size_t)v * NJ + j] != 0.f) inf.push_back({-W[(size_t)v * NJ + j], j});
        std::sort(inf.begin(), inf.end());
        for (int k = 0; k < 4; ++k) inf4[v][k] = k < (int)inf.size() ? inf[k].second : NJ;
      }
      // (a) lexicographic: (dominant joint, second, third, fourth, file index)
      std::vector<int> lex(order);
      std::sort(lex.begin(), lex.end(), [&](int a, int b) { return std::tie(inf4[a], a) < std::tie(inf4[b], b); });
      // (b) along the KINEMATIC CHAINS: body parts (dominant joint) in depth-first order of the skeleton, so that neighbouring
      //     parts share joints; inside a part the vertices also tied to the previous part first, those tied to the next part last,
      //     the rest by their other joints.  A tile that straddles two parts then sees few joints beyond either part's own
      //     (capsule body in a random file order: 0 of 216 tiles above 8 joints, mean 4.4 -- the lexicographic order: 4, mean 4.8)
      std::vector<int> dfs, pos(NJ + 1, NJ);
      {
        std::vector<int> stack{0};
        while (!stack.empty()) {
          const int j = stack.back(); stack.pop_back();
          dfs.push_back(j);
          for (int q = NJ - 1; q > j; --q) if (parents[q] == j) stack.push_back(q);      // children in ascending order
        }
        for (int i = 0; i < (int)dfs.size(); ++i) pos[dfs[i]] = i;
      }
      std::vector<std::array<int, 6>> ckey(V);
      for (int v = 0; v < V; ++v) {
        const int p0 = pos[inf4[v][0]];
        const int prev_j = p0 > 0 ? dfs[p0 - 1] : -1, next_j = p0 + 1 < (int)dfs.size() ? dfs[p0 + 1] : -1;
        bool has_prev = false, has_next = false;
        std::array<int, 3> sec{NJ, NJ, NJ};
        for (int k = 1; k < 4; ++k) {
          const int j = inf4[v][k];
          if (j == NJ) continue;
          has_prev = has_prev || j == prev_j; has_next = has_next || j == next_j;
          sec[k - 1] = pos[j];
        }
        std::sort(sec.begin(), sec.end());
        ckey[v] = {p0, (has_prev && !has_next) ? 0 : (has_next && !has_prev) ? 2 : 1, sec[0], sec[1], sec[2], v};
      }
      std::vector<int> chain(order);
      std::sort(chain.begin(), chain.end(), [&](int a, int b) { return ckey[a] < ckey[b]; });
      const long cost_lex = order_cost(lex), cost_chain = order_cost(chain);
      const std::vector<int>& best = cost_chain <= cost_lex ? chain : lex;
      if (force || std::min(cost_chain, cost_lex) < cost_file) { order = best; permuted = true; }
      // HINT (jrr_model_create_hinted): the vertices the caller's regressor will read -- its support -- are stored FIRST, packed into
      // as few tiles as the joint-sparse kernels like: groups of hinted vertices along the kinematic chains whose joints number at
      // most 8 (one pass of the 8-slot kernels), each group topped up to a full tile with other vertices of the same joints; the
      // rest follows in the order chosen above.  The iterations of JRR_FLAG_SUPPORT_TILES then run one tile per group instead of
      // up to one per support entry.  Dropped (no effect) when a group cannot be completed within the 16-joint window.
      if (n_hint > 0) {
        std::vector<char> is_hint(V, 0), used(V, 0);
        std::vector<int> hinted;
        for (int i = 0; i < n_hint; ++i) if (!is_hint[hint_vertices[i]]) { is_hint[hint_vertices[i]] = 1; hinted.push_back(hint_vertices[i]); }
        std::sort(hinted.begin(), hinted.end(), [&](int x, int y) { return ckey[x] < ckey[y]; });
        auto joints_of = [&](int v) { unsigned m_ = 0; for (int j = 0; j < NJ; ++j) if (W[(size_t)v * NJ + j] != 0.f) m_ |= 1u << j; return m_; };
        std::vector<int> with_hint;
        bool ok = true;
        size_t at = 0;
        while (at < hinted.size() && ok) {
          unsigned uni = 0;
          std::vector<int> tile;
          while (at < hinted.size() && tile.size() < 32) {      // the next group: joints <= 8 (a single vertex may bring up to 4)
            const unsigned grown = uni | joints_of(hinted[at]);
            if (!tile.empty() && __builtin_popcount(grown) > 8) break;
            uni = grown; tile.push_back(hinted[at]); used[hinted[at]] = 1; ++at;
          }
          for (int limit : {0, 8, KJS_TILE_MAX}) {               // fillers: joints inside the group's, then anything that keeps <= 8, <= 16
            for (int v : order) {
              if (tile.size() == 32) break;
              if (used[v] || is_hint[v]) continue;
              const unsigned grown = uni | joints_of(v);
              if (limit == 0 ? grown != uni : __builtin_popcount(grown) > limit) continue;
              uni = grown; tile.push_back(v); used[v] = 1;
            }
            if (tile.size() == 32) break;
          }
          ok = tile.size() == 32;
          with_hint.insert(with_hint.end(), tile.begin(), tile.end());
        }
        if (ok) {
          for (int v : order) if (!used[v]) with_hint.push_back(v);
          if ((int)with_hint.size() == V && order_cost(with_hint) < ((long)1 << 40)) { order = with_hint; permuted = true; hint_applied = (int)hinted.size(); }
        }
      }
    }
  }
  for (int p_ = 0; p_ < VP; ++p_) P2V[p_] = p_ < V ? order[p_] : -1;
  for (int p_ = 0; p_ < V; ++p_) V2P[order[p_]] = p_;
  for (int p_ = 0; p_ < V; ++p_) {
    const int v = order[p_];                 // vertex of the file stored in row p_
    const int t = p_ >> 5, vv = p_ & 31;
    for (int c = 0; c < 3; ++c) {
      for (int k = 0; k < KF; ++k) {
        float val;
        if (k < 207) val = pd[(size_t)k * (V * 3) + v * 3 + c];
        else if (k < 217) val = sd[((size_t)v * 3 + c) * NB + (k - 207)];
        else val = vt[v * 3 + c];
        Dk[((((size_t)t * (KFP / 4) + (k >> 2)) * 3 + c) * 32 + vv) * 4 + (k & 3)] = val;      // K-quads [tile][k / 4][plane][32 v][4]
        Dn[((size_t)c * VP + p_) * KFP + k] = val;
        Dq[(((size_t)c * (VP / 4) + (p_ >> 2)) * KFP + k) * 4 + (p_ & 3)] = val;
      }
    }
    for (int j = 0; j < NJ; ++j) {
      Wjv[((size_t)t * NJ + j) * 32 + vv] = W[(size_t)v * NJ + j];
      Wvj[((size_t)t * 32 + vv) * 32 + j] = W[(size_t)v * NJ + j];
    }
  }
  // joint-sparse skinning tables (jrr_common.h): per 32-vertex tile the joints with a non-zero weight.  PER-TILE classes: the
  // kernels are built for `kjs` (8 or 12) joint slots per tile and pass; a tile with more joints (up to KJS_TILE_MAX = 16:
  // the backward kernel's joint windows) costs ITSELF a second pass over slots kjs .. 2 kjs - 1, nobody else anything.
  int kjs = 0, wide_tiles = 0, most_joints = 0;
  int hist[NJ + 1] = {0};
  {
    std::vector<std::vector<int>> lists(VT);
    size_t most = 0;
    for (int t = 0; t < VT; ++t) {
      for (int j = 0; j < NJ; ++j) {
        bool used = false;
        for (int vv = 0; vv < 32 && !used; ++vv) { const int p_ = t * 32 + vv; used = p_ < V && W[(size_t)order[p_] * NJ + j] != 0.f; }
        if (used) lists[t].push_back(j);
      }
      most = std::max(most, lists[t].size());
      ++hist[lists[t].size()];
    }
    most_joints = (int)most;
    if (most <= (size_t)KJS_TILE_MAX) {
      // matrix instructions per tile of the forward kernel: 423 / 447 with 8 / 12 slots, + one pass (2 kjs x 3 + ~3 stage
      // hand-overs) for a wide tile
      long cost8 = 0, cost12 = 0;
      for (int t = 0; t < VT; ++t) {
        cost8 += 423 + (lists[t].size() > 8 ? 48 + 30 : 0);
        cost12 += 447 + (lists[t].size() > 12 ? 72 + 30 : 0);
      }
      kjs = cost8 <= cost12 ? 8 : KJS_MAX;
    }
    // JRR_DENSE_SKINNING=1 forces the dense kernels, JRR_SKIN_JOINTS=12 the 12-slot variant (verification: tests)
    { const char* dense = getenv("JRR_DENSE_SKINNING"); if (dense && dense[0] == '1') kjs = 0; }
    { const char* kj = getenv("JRR_SKIN_JOINTS"); if (kj && atoi(kj) == 12 && kjs == 8) kjs = 12; }
    for (int t = 0; t < VT && kjs; ++t) wide_tiles += (int)lists[t].size() > kjs;
    // segments for the backward kernel's 16-row dA windows: greedy runs of tiles whose joint union stays <= 16
    if (kjs) {
      int seg = 0;
      std::vector<int> win;      // joints of the current segment, in order of first appearance
      std::vector<int> first_tile{0};
      for (int t = 0; t < VT; ++t) {
        std::vector<int> grown(win);
        for (int j : lists[t]) if (std::find(grown.begin(), grown.end(), j) == grown.end()) grown.push_back(j);
        if (grown.size() > 16) {   // close the segment: its window is final
          for (int u = first_tile[seg]; u < t; ++u) for (int n = 0; n < 16; ++n) SegJ[u * 16 + n] = n < (int)win.size() ? win[n] : -1;
          ++seg; first_tile.push_back(t);
          win = lists[t];
        } else win = grown;
        SegId[t] = seg;
      }
      for (int u = first_tile[seg]; u < VT; ++u) for (int n = 0; n < 16; ++n) SegJ[u * 16 + n] = n < (int)win.size() ? win[n] : -1;
      for (int t = 0; t < VT; ++t)
        for (int n = 0; n < 16; ++n) {
          const int j = SegJ[t * 16 + n];
          for (int vv = 0; vv < 32; ++vv) {
            const int p_ = t * 32 + vv;
            W16[((size_t)t * 16 + n) * 36 + vv] = (j >= 0 && p_ < V) ? W[(size_t)order[p_] * NJ + j] : 0.f;
          }
        }
    }
    for (int t = 0; t < VT && kjs; ++t) {
      Tnj[t] = (int)lists[t].size();
      for (int n = 0; n < NJ; ++n) {                                   // NJ slots per tile: the tile's joints, ascending, then padding
        const int j = n < (int)lists[t].size() ? lists[t][n] : 0;      // padding: joint 0 with zero weights
        Jl[t * NJ + n] = j;
        for (int vv = 0; vv < 32; ++vv) {
          const int p_ = t * 32 + vv;
          Wc[((size_t)t * NJ + n) * 32 + vv] = (n < (int)lists[t].size() && p_ < V) ? W[(size_t)order[p_] * NJ + j] : 0.f;
        }
      }
    }
  }
  // folded rest-joint regressor: J(beta) = Jt + JS beta   (smplx vertices2joints(J_regressor, v_shaped))
  for (int j = 0; j < NJ; ++j)
    for (int c = 0; c < 3; ++c) {
      double acc = 0;
      double accs[NB] = {0};
      for (int v = 0; v < V; ++v) {
        const double w = Jr[(size_t)j * V + v];
        if (w == 0.0) continue;
        acc += w * vt[v * 3 + c];
        for (int l = 0; l < NB; ++l) accs[l] += w * sd[((size_t)v * 3 + c) * NB + l];
      }
      Jt[j * 3 + c] = (float)acc;
      for (int l = 0; l < NB; ++l) JS[(j * 3 + c) * NB + l] = (float)accs[l];
    }
  jrr_model* m = new jrr_model();
  void* base = buffer_dev;
  hipError_t e = hipSuccess;
  if (h.size() != model_floats()) { delete m; jrr_set_error("internal: model size mismatch"); return JRR_ERR_ARG; }
  if (!base) {      // no caller buffer: the library allocates (and frees) its own
    e = hipMalloc(&base, jrr_model_bytes());
    if (e != hipSuccess) { delete m; jrr_set_error("hipMalloc(model) failed: %s", hipGetErrorString(e)); return JRR_ERR_HIP; }
  }
  e = hipMemcpy(base, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) { if (!buffer_dev) (void)hipFree(base); delete m; jrr_set_error("hipMemcpy(model) failed: %s", hipGetErrorString(e)); return JRR_ERR_HIP; }
  float* d = (float*)base;
  m->base = base;
  m->owns_base = buffer_dev == nullptr;
  m->faces_area = reinterpret_cast<int*>((char*)base + round_up(model_floats() * sizeof(float), 256));
  m->d.Dk = d;
  m->d.Dn = d + nDk;
  m->d.Dq = m->d.Dn + nDn;
  m->d.Wjv = m->d.Dq + nDq;
  m->d.Wvj = m->d.Wjv + nWjv;
  m->d.Jt = m->d.Wvj + nWvj;
  m->d.JS = m->d.Jt + nJt;
  m->d.Wc = m->d.JS + nJS;
  m->d.jl = reinterpret_cast<int*>(m->d.Wc + nWc);
  m->d.kjs = kjs;
  m->d.tnj = m->d.jl + (size_t)VT * NJ;
  m->d.wide_tiles = wide_tiles;
  m->d.most_joints = most_joints;
  for (int n = 0; n <= NJ; ++n) m->d.tile_hist[n] = hist[n];
  m->d.permuted = permuted ? 1 : 0;
  m->d.hint_applied = hint_applied;
  m->v2p_host = nullptr;
  if (permuted) { m->v2p_host = new int[V]; memcpy(m->v2p_host, V2P, (size_t)V * sizeof(int)); }
  m->d.p2v = permuted ? reinterpret_cast<int*>(m->d.jl + nJl) : nullptr;
  m->d.v2p = permuted ? m->d.p2v + VP : nullptr;
  m->d.W16 = reinterpret_cast<float*>(reinterpret_cast<int*>(m->d.jl + nJl) + nPerm);
  m->d.segid = reinterpret_cast<int*>(m->d.W16 + nW16);
  m->d.segj = m->d.segid + (size_t)VT * 16;
  { const char* b16 = getenv("JRR_BWD16"); m->d.bwd16 = (kjs && !(b16 && b16[0] == '0')) ? 1 : 0; }
  m->d.role_kjs = (kjs && !wide_tiles) ? kjs : 0;
  m->d.parents.maxd = 0;
  m->d.faces = nullptr;
  m->d.faces_int = nullptr;
  m->d.faces_pk = nullptr;
  m->d.faces_int_pk = nullptr;
  m->d.nfaces = 0;
  for (int j = 0; j < NJ; ++j) {
    m->d.parents.p[j] = parents[j];
    m->d.parents.depth[j] = (j == 0) ? 0 : m->d.parents.depth[parents[j]] + 1;
    if (m->d.parents.depth[j] > m->d.parents.maxd) m->d.parents.maxd = m->d.parents.depth[j];
  }
  {
    int n = 0;
    for (int j = 0; j < NJ; ++j) {
      m->d.parents.child_off[j] = (unsigned char)n;
      for (int q = j + 1; q < NJ; ++q)
        if (parents[q] == j) m->d.parents.child[n++] = (unsigned char)q;
    }
    m->d.parents.child_off[NJ] = (unsigned char)n;
    for (; n < NJ; ++n) m->d.parents.child[n] = 0;
  }
  *out = m;
  return JRR_OK;
}

extern "C" int jrr_model_set_faces(jrr_model_t* m, const int32_t* faces, int n_faces) {
  if (!m || !faces || n_faces <= 0) return JRR_ERR_ARG;
  if (n_faces > MAX_FACES) { jrr_set_error("%d faces: the rasteriser holds at most %d", n_faces, MAX_FACES); return JRR_ERR_ARG; }
  for (int i = 0; i < n_faces * 3; ++i)
    if (faces[i] < 0 || faces[i] >= V) { jrr_set_error("face index %d out of range", faces[i]); return JRR_ERR_ARG; }
  // both index lists live in the tail of the model buffer (jrr_model_bytes): no allocation here
  m->d.faces = m->faces_area;
  m->d.faces_int = nullptr;
  JRR_HIP(hipMemcpy(m->d.faces, faces, (size_t)n_faces * 3 * sizeof(int), hipMemcpyHostToDevice));
  if (m->v2p_host) {      // the fused rasteriser reads the vertices in the internal order: faces in row indices
    std::vector<int32_t> fi((size_t)n_faces * 3);
    for (size_t i = 0; i < fi.size(); ++i) fi[i] = m->v2p_host[faces[i]];
    m->d.faces_int = m->faces_area + (size_t)MAX_FACES * 3;
    JRR_HIP(hipMemcpy(m->d.faces_int, fi.data(), fi.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  // the rasteriser reads a face as ONE 8-byte record (three 13-bit vertex indices): one gather per face where three strided
  // 4-byte ones were the resolve pass's bound
  auto pack = [&](const int32_t* f3, unsigned* dst) -> int {
    std::vector<unsigned> pk((size_t)n_faces * 2);
    for (int i = 0; i < n_faces; ++i) { pk[2 * i] = (unsigned)f3[3 * i] | ((unsigned)f3[3 * i + 1] << 13); pk[2 * i + 1] = (unsigned)f3[3 * i + 2]; }
    return hipMemcpy(dst, pk.data(), pk.size() * sizeof(unsigned), hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
  };
  static_assert(V <= 8192, "packed face records hold 13-bit vertex indices");
  m->d.faces_pk = reinterpret_cast<unsigned*>(m->faces_area + (size_t)2 * MAX_FACES * 3);
  m->d.faces_int_pk = nullptr;
  if (pack(faces, m->d.faces_pk)) { jrr_set_error("hipMemcpy(faces) failed"); return JRR_ERR_HIP; }
  if (m->v2p_host) {
    std::vector<int32_t> fi((size_t)n_faces * 3);
    for (size_t i = 0; i < fi.size(); ++i) fi[i] = m->v2p_host[faces[i]];
    m->d.faces_int_pk = m->d.faces_pk + (size_t)MAX_FACES * 2;
    if (pack(fi.data(), m->d.faces_int_pk)) { jrr_set_error("hipMemcpy(faces) failed"); return JRR_ERR_HIP; }
  }
  m->d.nfaces = n_faces;
  return JRR_OK;
}

extern "C" void jrr_model_destroy(jrr_model_t* m) {
  if (!m) return;
  if (m->base && m->owns_base) (void)hipFree(m->base);
  delete[] m->v2p_host;
  delete m;
}

extern "C" int jrr_model_info(const jrr_model_t* m, int32_t* out, int n) {
  if (!m || !out) return JRR_ERR_ARG;
  int32_t v[4 + NJ + 2] = {m->d.kjs, m->d.wide_tiles, m->d.most_joints, m->d.permuted};
  for (int k = 0; k <= NJ; ++k) v[4 + k] = m->d.tile_hist[k];
  v[4 + NJ + 1] = m->d.hint_applied;
  for (int i = 0; i < n && i < 4 + NJ + 2; ++i) out[i] = v[i];
  return JRR_OK;
}

// =============================================================================================
// engine
// =============================================================================================
struct jrr_engine {
  Model m;
  int B, BP, bnorm, flags;
  int sil;                                           // silhouette image size: 224, or 256 with JRR_FLAG_SIL_256
  int nvc, nvcb, nsplit, nsplitJ;
  bool have_J, have_mask, have_pd, have_sd, has_model;
  // workspace sections
  float *rowsum, *Jraw, *Jmask, *Jn, *Jn_vi, *Jn_iv, *Jn_q;
  bool tab_static;                                   // the W parts of the backward operand records are in place
  float *FT, *FTq, *AT, *VPb, *JP, *dJT, *DVP, *dATp, *dFTp, *joints, *sqerr, *Jsum, *dA, *dF, *R0T, *dRT, *dbT;
  unsigned* dmask;                                   // [slab][BP / 64] joint masks of the dA slabs (k_lbs_bwd16 -> k_chain_bwd)
  float *convL;                                      // LDS image of the per-joint MLP parameters (k_conv_image)
  float *W2s, *zpart;                                // fc2.w rows scaled by fc4.w; partial fc4 dots [16][BP]
  float *W0Tq, *W2Tq, *W2sq, *W0q;                   // the four GEMM weight operands in quads [k/4][m][4]
  float *Pd, *W0T, *W2T, *H2T, *A1T, *A2T, *dA2T, *dA1T, *dH2T, *gx, *TrA, *TrB, *dz0, *dsc, *wgs;
  float *Ps, *gb;
  float *dsq, *ssq;                                  // per-pose squared adversarial errors of the last iteration [25][BP], [BP]
  long long* probe;                                  // shader-clock probe of k_lbs_fwd (profiling)
  float *ndc, *sqsil, *VPM; unsigned* cover; int* ncover;   // soft silhouette (JRR_FLAG_SILHOUETTE)
  const float* sil_mask; float* smask; bool smask_valid;     // target masks, per-pose sum(mask^2)
  float *JW, *Hm, *Hk, *G0, *MT, *dMT;      // folded regressor (JRR_FLAG_FOLDED)
  float* Dsplit;                            // the blend basis as bf16 hi | lo chunks (JRR_FLAG_BLEND_BF16X3)
  bool folded, fold_valid;
  // forward reuse (jrr_refine_run_after_j_step): state left by jrr_j_regressor_grad's SMPL forward; dropped by every
  // entry point that overwrites FT / AT / VPb / VTb or may run between the two calls (drop_cached_forward)
  bool fwd_cached; const float *fc_x6d, *fc_betas;
  float* dJraw;                                      // (17,6890) gradient scratch of the in-call J steps (jrr_refine_run_j_steps)
  JSupport jsup; bool have_jsup;                     // support lists of the normalised regressor (KEEP_VERTS engines; lbs.hip)
  bool jsup_fits_known;                              // jrr_j_support_info has seen flag = 1 for the current regressor lineage
  const float* jsup_mask;                            // ... under this mask (another mask may un-mask entries: knowledge dropped)
  float* hist; int hist_cap, hist_every, hist_n; long long hist_iter;   // loss history (jrr_engine_set_loss_history)
  float *VTb;       // [3][VP][BP] vertices / transposed vertex adjoint (KEEP_VERTS or SILHOUETTE)
  float *dVTb, *dJnp, *dJn;   // transposed external vertex adjoint [3][VP][BP]; J-gradient partial slabs [3*nsplitJ][32][VP]
  int32_t* step_scratch;
  bool profiling;
  bool verts_partial;                                // VTb holds the support tiles of the last J step only
  // JRR_FLAG_SUPPORT_TILES: the 32-vertex tiles that hold an entry of the regressor's support (ascending), taken when
  // jrr_j_support_info reports that the support fits; J steps only shrink the support, so the list stays a superset
  int* act_list; int nact; bool act_valid;
  // ... and, when the support has at most SUP_NSV vertices, the joint-loss iteration runs per VERTEX in one workgroup per 32-pose group
  // (supk.h): the gathered basis rows and skinning lists of the support, built with the tile list
  SupTables sup; int sup_nsv; bool sup_valid;
  // JRR_SUP_OVERLAP: a second stream for the half of the support-vertex iteration that does not read the discriminator GEMMs' results,
  // and the two events that fork / join it (created on first use)
  hipStream_t side; hipEvent_t ev_fork, ev_join;
  std::vector<hipEvent_t>* ev[JRR_PROF_CLASSES];
  const float* gt_j2d; float* cam; float* cam_m; float* cam_v;   // 2-D reprojection term (nullable)
  float *gcam, *sq2d;
};

// number of vertex chunks such that the grid fills whole "rounds" of the chip's resident
// workgroup slots (256 CUs x 2 workgroups): a grid of 1.1 rounds costs 2 rounds of time.
static int pick_chunks(int wg_per_chunk, int max_chunks) {
  const int slots = 512;
  int best = 1;
  double best_score = -1.0;
  for (int n = 1; n <= max_chunks; ++n) {
    const int total = wg_per_chunk * n;
    const int rounds = (total + slots - 1) / slots;
    const double fill = (double)total / ((double)rounds * slots);
    // tiles per chunk differ by at most one: the longest chunk sets the time of its round
    const double tiles = 216.0 / n;
    const double balance = tiles / (double)((216 + n - 1) / n);
    const double score = fill * balance - 0.0002 * n;   // prefer fewer chunks at equal efficiency
    if (score > best_score) { best_score = score; best = n; }
  }
  return best;
}

static void plan_geometry(int BP, int& nvc, int& nvcb, int& nvcb16, int& nsplit, int& nsplitJ) {
  const int nbg = BP / BG;
  // forward: one workgroup per (128 poses, chunk).  At most 54 chunks (4 tiles each) -- 108 for batches of up to 256 poses (round 6), whose
  // one or two pose groups would otherwise put 54 / 108 workgroups on 512 slots with 4 tiles to walk each (256 poses, the reference's
  // default batch, all five terms: k_lbs_fwd 67 -> 40 us, k_joints_loss 12 -> 17 us for twice the partials, the iteration 0.390 -> 0.367 ms;
  // 216 chunks: 0.378; at 512 poses 108 chunks are two workgroups per CU and gain nothing: tools/exp/fwd_chunk_cap_ab.sh).
  // JRR_FWD_CHUNK_CAP (experiments) overrides the cap.
  int fwd_cap = nbg <= 2 ? 108 : 54;
  { const char* e = getenv("JRR_FWD_CHUNK_CAP"); if (e && atoi(e) >= 1 && atoi(e) <= 216) fwd_cap = atoi(e); }
  nvc = pick_chunks(nbg, fwd_cap);
  // exactly one round of 512 workgroups with an even chunk count lets k_lbs_fwd pair the two workgroups of a CU (lbs.hip)
  { const char* e = getenv("JRR_FWD_ROUND"); if (!(e && e[0] == '0') && 512 % nbg == 0 && 512 / nbg <= 64 && ((512 / nbg) & 1) == 0 && 32 % (512 / nbg / 2) == 0) nvc = 512 / nbg; }
  nvcb = pick_chunks(BP / BT, 36);                  // backward, role kernel: one workgroup per (32 poses, chunk)
  nvcb16 = pick_chunks(BP / 64, 36);                // backward, k_lbs_bwd16: one workgroup per (64 poses, chunk)
  // ... but exactly one round of 512 workgroups with an even chunk count lets the kernel pair the two workgroups of a CU on
  // one pose group (lbs.hip), which is worth more than the last per cent of tile balance
  if (512 % (BP / 64) == 0 && 512 / (BP / 64) <= 36 && ((512 / (BP / 64)) & 1) == 0 && 32 % (512 / (BP / 64) / 2) == 0) nvcb16 = 512 / (BP / 64);
  nsplit = (512 + nbg - 1) / nbg;
  // at most 32 slabs -- 64 for batches of up to 512 poses, whose 32 x (BP / 128) workgroups leave most of the chip idle (256 poses, the
  // reference's default batch: blend adjoint 81 -> 46 us, the iteration 0.383 -> 0.357 ms; at 1024 poses 64 slabs gain the product
  // nothing and cost the slab sum 10 us: DESIGN.md section 8)
  const int cap = nbg <= 4 ? 64 : 32;
  if (nsplit > cap) nsplit = cap;
  if (nsplit < 1) nsplit = 1;
  // (experiment knobs, tools/exp: the split-K slab count of the blend adjoint and the vertex chunks of k_lbs_bwd16)
  { const char* e = getenv("JRR_NSPLIT"); if (e && atoi(e) >= 1 && atoi(e) <= 256) nsplit = atoi(e); }
  { const char* e = getenv("JRR_NVCB16"); if (e && atoi(e) >= 1 && atoi(e) <= 36) nvcb16 = atoi(e); }
  nsplitJ = 3;                                      // pose splits per plane of the J-gradient product: 162 x 3 workgroups
  if (BP / 32 < nsplitJ) nsplitJ = BP / 32;         // = 0.95 of one round of the chip's 512 workgroup slots
}

struct Carver {
  char* base; size_t off;
  float* take(size_t nfloats) {
    float* p = base ? (float*)(base + off) : nullptr;
    off += round_up(nfloats * sizeof(float), 256);
    return p;
  }
};

// silhouette image size of an engine: JRR_FLAG_SIL_SIZE(size) if given, else 256 with JRR_FLAG_SIL_256, else 224
static int sil_size_of(int flags) {
  const int k = (flags & JRR_FLAG_SIL_SIZE_MASK) >> 16;
  return k ? 32 * k : (flags & JRR_FLAG_SIL_256) ? 256 : 224;
}

static size_t carve(jrr_engine* e, void* ws, int B, int flags) {
  const int BP = (int)round_up((size_t)B, BG);
  int nvc, nvcb, nvcb16, nsplit, nsplitJ;
  plan_geometry(BP, nvc, nvcb, nvcb16, nsplit, nsplitJ);
  Carver c{(char*)ws, 0};
  jrr_engine tmp;
  jrr_engine* t = e ? e : &tmp;
  t->rowsum = c.take(32);
  t->probe = (long long*)c.take(16);   // 3 x int64 used
  if (!(flags & JRR_FLAG_NO_MODEL)) {  // SMPL sections (~230 KB per pose): not carved for a discriminator-only engine
    t->Jraw = c.take((size_t)NH * V);
    t->Jmask = c.take((size_t)NH * V);
    t->Jn = c.take((size_t)NH * V);
    t->Jn_vi = c.take((size_t)VT * 1024);
    t->Jn_iv = c.take((size_t)VT * 2560);   // backward per-tile operand records [Jn | W^T | W | pad]
    t->Jn_q = c.take((size_t)VP * 32);      // normalised regressor in vertex quads [VP/4][32][4]
    t->FT = c.take((size_t)KFP * BP);
    t->FTq = c.take((size_t)KFP * BP);      // the features in K-quads: B operand of k_lbs_fwd's blend product
    t->AT = c.take((size_t)12 * NJ * BP);
    t->VPb = c.take((size_t)3 * VP * BP);
    t->JP = c.take((size_t)nvc * 3 * NH * BP);
    t->dJT = c.take((size_t)3 * NHP * BP);
    t->DVP = c.take((size_t)3 * VP * BP);
    t->dATp = c.take((size_t)(nvcb > nvcb16 ? nvcb : nvcb16) * 12 * NJ * BP);     // slabs of either backward kernel
    t->dmask = (unsigned*)c.take((size_t)(nvcb > nvcb16 ? nvcb : nvcb16) * (BP / 64));   // joints present in each slab (k_lbs_bwd16)
    t->dFTp = c.take((size_t)nsplit * KFP * BP);
    t->Jsum = c.take((size_t)3 * NH * BP);
    t->dA = c.take((size_t)12 * NJ * BP);
    t->dF = c.take((size_t)KFP * BP);
    t->R0T = c.take((size_t)16 * BP);
    t->dRT = c.take((size_t)NJ * 9 * BP);
    t->dbT = c.take((size_t)16 * BP);
  }
  t->joints = c.take((size_t)BP * NH * 3);
  t->sqerr = c.take((size_t)BP);
  t->step_scratch = (int32_t*)c.take(64);
  t->gcam = c.take((size_t)BP * 3);
  t->sq2d = c.take((size_t)BP);
  if (flags & JRR_FLAG_POSE_DISC) {
    t->Pd = c.take(DP_TOTAL);
    t->W0T = c.take((size_t)768 * 1024);
    t->W2T = c.take((size_t)1024 * 1024);
    t->W2s = c.take((size_t)1024 * 1024);
    t->W0Tq = c.take((size_t)768 * 1024);
    t->W2Tq = c.take((size_t)1024 * 1024);
    t->W2sq = c.take((size_t)1024 * 1024);
    t->W0q = c.take((size_t)1024 * 768);
    t->zpart = c.take((size_t)16 * BP);
    t->convL = c.take(CONV_IMAGE_FLOATS);
    t->H2T = c.take((size_t)768 * BP);
    t->A1T = c.take((size_t)1024 * BP);
    t->A2T = c.take((size_t)1024 * BP);
    t->dA2T = c.take((size_t)1024 * BP);
    t->dA1T = c.take((size_t)1024 * BP);
    t->dH2T = c.take((size_t)768 * BP);
    t->gx = c.take((size_t)BP * JRR_POSE6D);
    t->TrA = c.take((size_t)BP * 1024);
    t->TrB = c.take((size_t)BP * 1024);
    {   // weight-gradient partial slabs: 8 pose-splits of a 1024x1024 layer, or one conv/head slab per wave
      const size_t conv = (size_t)(BP / 64) * (NJ * 1280 + 792) + (size_t)NJ * 1280, fc = (size_t)8 * 1024 * 1024;
      t->wgs = c.take(conv > fc ? conv : fc);
    }
    t->dz0 = c.take((size_t)BP);
    t->dsc = c.take((size_t)BP * 25);
    t->dsq = c.take((size_t)BP * 25);
  }
  if (flags & JRR_FLAG_SHAPE_DISC) {
    t->Ps = c.take(256);
    t->gb = c.take((size_t)BP * NB);
    t->ssq = c.take((size_t)BP);
  }
  if (flags & (JRR_FLAG_SILHOUETTE | JRR_FLAG_KEEP_VERTS)) t->VTb = c.take((size_t)3 * VP * BP);
  if (flags & JRR_FLAG_SILHOUETTE) {
    t->ndc = c.take((size_t)BP * V * 4);
    const size_t S = (size_t)sil_size_of(flags);
    t->cover = (unsigned*)c.take((size_t)BP * S * S);
    t->ncover = (int*)c.take((size_t)BP);
    t->sqsil = c.take((size_t)BP);
    t->smask = c.take((size_t)BP);
    t->VPM = c.take((size_t)BP * 3 * VP);      // the vertices of a silhouette iteration, pose-major (k_lbs_fwd -> fused rasteriser)
  }
  if (flags & JRR_FLAG_FOLDED) {
    t->JW = c.take((size_t)VP * FOLD_MJ);
    t->Hm = c.take((size_t)FOLD_M * KFP);
    t->Hk = c.take((size_t)KFP * FOLD_M);
    t->G0 = c.take(FOLD_MJ);
    t->MT = c.take((size_t)FOLD_M * BP);
    t->dMT = c.take((size_t)FOLD_M * BP);
  }
  if ((flags & JRR_FLAG_BLEND_BF16X3) && !(flags & JRR_FLAG_NO_MODEL)) t->Dsplit = c.take(blend_basis_split_bytes() / sizeof(float));
  if (flags & JRR_FLAG_KEEP_VERTS) {
    t->dVTb = c.take((size_t)3 * VP * BP);
    t->dJnp = c.take((size_t)3 * nsplitJ * 32 * VP);
    t->dJn = c.take((size_t)NH * VP);
    t->dJraw = c.take((size_t)NH * V);
    t->jsup.flag = (int*)c.take(64);
    t->jsup.cnt = (int*)c.take(64);
    t->jsup.col = (int*)c.take((size_t)NH * JSUP_CAP);
    t->jsup.val = c.take((size_t)NH * JSUP_CAP);
    t->jsup.tmask = (int*)c.take(256);
    t->jsup.tknown = (int*)c.take(256);
    t->act_list = (int*)c.take(256);
    if (flags & JRR_FLAG_SUPPORT_TILES) { float* sb = c.take(sup_tables_floats()); if (sb) sup_tables_carve(t->sup, sb); }
  }
  if (e) {
    e->BP = BP; e->nvc = nvc; e->nvcb = (e->has_model && e->m.kjs && e->m.bwd16) ? nvcb16 : nvcb; e->nsplit = nsplit; e->nsplitJ = nsplitJ;
  }
  return c.off;
}

extern "C" size_t jrr_engine_workspace_bytes(int batch, int flags) {
  if (batch <= 0) return 0;
  return carve(nullptr, nullptr, batch, flags);
}

extern "C" int jrr_engine_create(const jrr_model_t* model, int batch, int batch_norm, void* ws, size_t ws_bytes,
                                 int flags, jrr_engine_t** out) {
  if (!ws || !out || batch <= 0) { jrr_set_error("jrr_engine_create: bad argument"); return JRR_ERR_ARG; }
  if ((flags & (JRR_FLAG_SIL_256 | JRR_FLAG_SIL_SIZE_MASK)) && !(flags & JRR_FLAG_SILHOUETTE)) { jrr_set_error("jrr_engine_create: a silhouette size without JRR_FLAG_SILHOUETTE"); return JRR_ERR_ARG; }
  if ((flags & JRR_FLAG_SIL_SIZE_MASK) && ((flags & JRR_FLAG_SIL_256) || ((flags & JRR_FLAG_SIL_SIZE_MASK) >> 16) > 8)) {
    jrr_set_error("jrr_engine_create: JRR_FLAG_SIL_SIZE takes a multiple of 32 up to 256 (and excludes JRR_FLAG_SIL_256)");
    return JRR_ERR_ARG;
  }
  if (!model && (flags & ~(JRR_FLAG_POSE_DISC | JRR_FLAG_SHAPE_DISC | JRR_FLAG_NO_MODEL))) {
    jrr_set_error("jrr_engine_create: a model-less engine serves the discriminators only");
    return JRR_ERR_ARG;
  }
  if (model && (flags & JRR_FLAG_NO_MODEL)) { jrr_set_error("jrr_engine_create: JRR_FLAG_NO_MODEL with a model"); return JRR_ERR_ARG; }
  if ((flags & JRR_FLAG_SUPPORT_TILES) && !(flags & JRR_FLAG_KEEP_VERTS)) { jrr_set_error("jrr_engine_create: JRR_FLAG_SUPPORT_TILES needs JRR_FLAG_KEEP_VERTS (the support lists live there)"); return JRR_ERR_ARG; }
  if (((uintptr_t)ws & 255) != 0) { jrr_set_error("workspace must be 256-byte aligned"); return JRR_ERR_ARG; }
  const size_t need = jrr_engine_workspace_bytes(batch, flags);
  if (ws_bytes < need) { jrr_set_error("workspace too small: %zu < %zu", ws_bytes, need); return JRR_ERR_WORKSPACE; }
  jrr_engine* e = new jrr_engine();
  memset((void*)e, 0, sizeof(*e));
  if (model) e->m = model->d;
  e->has_model = model != nullptr;
  e->B = batch;
  e->bnorm = batch_norm > 0 ? batch_norm : batch;
  e->flags = flags;
  e->sil = sil_size_of(flags);
  carve(e, ws, batch, flags);
  e->have_jsup = (flags & JRR_FLAG_KEEP_VERTS) != 0;
  if (flags & JRR_FLAG_BLEND_BF16X3) {      // side mode: the basis is split once, here
    launch_split_blend_basis(e->m.Dq, e->Dsplit, nullptr);
    const hipError_t he = hipStreamSynchronize(nullptr);
    if (he != hipSuccess) { jrr_set_error("jrr_engine_create: splitting the blend basis failed: %s", hipGetErrorString(he)); delete e; return JRR_ERR_HIP; }
  }
  *out = e;
  return JRR_OK;
}

static void clear_events(jrr_engine* e) {
  for (int c = 0; c < JRR_PROF_CLASSES; ++c) {
    if (!e->ev[c]) continue;
    for (hipEvent_t ev : *e->ev[c]) (void)hipEventDestroy(ev);
    e->ev[c]->clear();
  }
}

extern "C" void jrr_engine_destroy(jrr_engine_t* e) {
  if (!e) return;
  clear_events(e);
  for (int c = 0; c < JRR_PROF_CLASSES; ++c) delete e->ev[c];
  if (e->side) { (void)hipStreamSynchronize(e->side); (void)hipStreamDestroy(e->side); (void)hipEventDestroy(e->ev_fork); (void)hipEventDestroy(e->ev_join); }
  delete e;
}

extern "C" int jrr_engine_set_profiling(jrr_engine_t* e, int enabled) {
  if (!e) return JRR_ERR_ARG;
  e->profiling = enabled != 0;
  for (int c = 0; c < JRR_PROF_CLASSES; ++c)
    if (!e->ev[c]) e->ev[c] = new std::vector<hipEvent_t>();
  if (!e->profiling) clear_events(e);
  return JRR_OK;
}

extern "C" int jrr_engine_probe_read(jrr_engine_t* e, int64_t* out_host) {
  if (!e || !out_host) return JRR_ERR_ARG;
  long long v[3] = {0, 0, 0};
  JRR_HIP(hipMemcpy(v, e->probe, sizeof(v), hipMemcpyDeviceToHost));
  out_host[0] = v[0]; out_host[1] = v[1];
  out_host[2] = 2;          // waves resident per SIMD (2 workgroups of 4 waves per CU: launch bounds + 70 KB LDS)
  out_host[3] = 16 * 4;     // issue clocks per v_mfma_f32_32x32x2_f32 (16 passes of 4 clocks)
  out_host[4] = v[2] * 10;  // the interval of out[0] in ns (s_memrealtime, 100 MHz)
  return JRR_OK;
}

extern "C" int jrr_engine_profile_read(jrr_engine_t* e, float* ms_host, int32_t* counts_host) {
  if (!e || !ms_host) return JRR_ERR_ARG;
  for (int c = 0; c < JRR_PROF_CLASSES; ++c) {
    double tot = 0;
    int n = 0;
    if (e->ev[c]) {
      for (size_t i = 0; i + 1 < e->ev[c]->size(); i += 2) {
        float ms = 0.f;
        JRR_HIP(hipEventSynchronize((*e->ev[c])[i + 1]));
        JRR_HIP(hipEventElapsedTime(&ms, (*e->ev[c])[i], (*e->ev[c])[i + 1]));
        tot += ms;
        ++n;
      }
    }
    ms_host[c] = n ? (float)(tot / n) : 0.f;
    if (counts_host) counts_host[c] = n;
  }
  clear_events(e);
  return JRR_OK;
}

// RAII-less bracket helper: records an event on the stream if profiling is on
static inline void prof_mark(jrr_engine* e, int cls, hipStream_t s) {
  if (!e->profiling) return;
  hipEvent_t ev;
  if (hipEventCreate(&ev) != hipSuccess) return;
  (void)hipEventRecord(ev, s);
  e->ev[cls]->push_back(ev);
}

extern "C" int jrr_engine_set_batch_norm(jrr_engine_t* e, int bn) {
  if (!e || bn <= 0) return JRR_ERR_ARG;
  e->bnorm = bn;
  return JRR_OK;
}

extern "C" int jrr_engine_info(const jrr_engine_t* e, int32_t* out, int n) {
  if (!e || !out) return JRR_ERR_ARG;
  int32_t v[9] = {e->B, e->BP, e->bnorm, e->nvc, e->nvcb, e->nsplit, e->nsplitJ, e->flags, e->has_model ? e->m.kjs : 0};
  for (int i = 0; i < n && i < 9; ++i) out[i] = v[i];
  return JRR_OK;
}

// H[(i,j,c)][k] = sum_v Jn[i,v] W[v,j] D_c[k,v]  and  G0 (fold.hip); both layouts of H
static int fold_rebuild(jrr_engine* e, hipStream_t s) {
  launch_fold_jw(e->Jn, e->m.Wjv, e->JW, e->G0, e->m.p2v, s);
  for (int c = 0; c < 3; ++c) {
    GemmArgs g;
    g.A = e->JW; g.lda = FOLD_MJ;                          // A[k = v][m = (i,j)]
    g.Bm = e->m.Dn + (size_t)c * VP * KFP; g.ldb = KFP;    // Bm[k = v][n = feature]
    g.Out = e->Hm + (size_t)c * KFP; g.ldo = 3 * KFP;      // row (i,j) of plane c = row (i*24+j)*3 + c of Hm
    g.bias = nullptr; g.mask = nullptr; g.split_stride = 0;
    g.M = NH * NJ; g.N = KFP; g.K = VP;
    int rc = launch_gemm_128x32(g, EPI_STORE, 1, s);
    if (rc) return rc;
  }
  launch_transpose(e->Hm, e->Hk, NH * NJ * 3, KFP, s, KFP, FOLD_M);
  e->fold_valid = true;
  return 0;
}

extern "C" int jrr_engine_set_folded(jrr_engine_t* e, int enabled, void* stream) {
  if (!e) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (enabled && !(e->flags & JRR_FLAG_FOLDED)) { jrr_set_error("engine created without JRR_FLAG_FOLDED"); return JRR_ERR_STATE; }
  e->folded = enabled != 0;
  if (e->folded && e->have_J && !e->fold_valid) {
    int rc = fold_rebuild(e, (hipStream_t)stream);
    if (rc) return rc;
    CHECK_LAUNCH();
  }
  return JRR_OK;
}

static int set_j_regressor_impl(jrr_engine_t* e, const float* J, const float* mask, void* stream, int32_t* step_inc);
extern "C" int jrr_engine_set_j_regressor(jrr_engine_t* e, const float* J, const float* mask, void* stream) {
  return set_j_regressor_impl(e, J, mask, stream, nullptr);
}
// step_inc (J step only): the Adam step counter, incremented by the first launch of the normalisation
static int set_j_regressor_impl(jrr_engine_t* e, const float* J, const float* mask, void* stream, int32_t* step_inc) {
  if (!e || !J) { jrr_set_error("set_j_regressor: null"); return JRR_ERR_ARG; }
  if (!e->has_model) { jrr_set_error("engine was created without an SMPL model (discriminators only)"); return JRR_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  e->jsup_fits_known = false;      // a regressor from outside: its support is not known to fit until jrr_j_support_info says so
  if (e->verts_partial) e->fwd_cached = false;      // the stored vertices cover the OLD regressor's support tiles only
  e->act_valid = false; e->sup_valid = false;
  if (e->have_jsup && !step_inc) JRR_HIP(hipMemsetAsync(e->jsup.flag + JSUP_KNOWN, 0, sizeof(int32_t), s));   // (a J step keeps the baseline: step_inc != NULL)
  JRR_HIP(hipMemcpyAsync(e->Jraw, J, (size_t)NH * V * 4, hipMemcpyDeviceToDevice, s));
  if (mask) JRR_HIP(hipMemcpyAsync(e->Jmask, mask, (size_t)NH * V * 4, hipMemcpyDeviceToDevice, s));
  e->have_mask = mask != nullptr;
  if (mask != e->jsup_mask) e->jsup_fits_known = false;      // a different mask may un-mask entries: the support may have grown
  e->jsup_mask = mask;
  if (!e->tab_static) {                                                                     // model-only: once per engine
    launch_bwd_tab_static(e->m, e->Jn_iv, s); e->tab_static = true;
    // the support-restricted J step writes dJn on the support only: everything else must be finite (it meets Jn = 0)
    if (e->have_jsup) JRR_HIP(hipMemsetAsync(e->dJn, 0, (size_t)NH * VP * sizeof(float), s));
  }
  launch_jreg_normalize(e->Jraw, e->have_mask ? e->Jmask : nullptr, e->rowsum, e->Jn, e->Jn_vi, e->Jn_iv, e->Jn_q, e->m.p2v, s,
                        (e->m.kjs && e->m.bwd16) ? 1 : 0, e->m.v2p, e->have_jsup ? &e->jsup : nullptr, step_inc);
  e->fold_valid = false;
  if (e->folded) {
    int rc = fold_rebuild(e, s);
    if (rc) return rc;
  }
  CHECK_LAUNCH();
  e->have_J = true;
  return JRR_OK;
}

extern "C" int jrr_engine_set_pose_disc(jrr_engine_t* e, const float* P, void* stream) {
  if (!e || !P) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (!(e->flags & JRR_FLAG_POSE_DISC)) { jrr_set_error("engine created without JRR_FLAG_POSE_DISC"); return JRR_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  JRR_HIP(hipMemcpyAsync(e->Pd, P, (size_t)DP_TOTAL * 4, hipMemcpyDeviceToDevice, s));
  launch_transpose(e->Pd + DP_FC0_W, e->W0T, 1024, 768, s);    // [out][in] -> [in][out]
  launch_transpose(e->Pd + DP_FC2_W, e->W2T, 1024, 1024, s);
  launch_scale_rows(e->Pd + DP_FC2_W, e->Pd + DP_FC4_W, e->W2s, 1024, 1024, s);   // row n of fc2.w times fc4.w[n]
  launch_to_quads(e->W0T, 1024, e->W0Tq, 768, 1024, s);                // A operands A[k][m] of the four loop GEMMs, in quads
  launch_to_quads(e->W2T, 1024, e->W2Tq, 1024, 1024, s);
  launch_to_quads(e->W2s, 1024, e->W2sq, 1024, 1024, s);
  launch_to_quads(e->Pd + DP_FC0_W, 768, e->W0q, 1024, 768, s);
  launch_conv_image(e->Pd, e->convL, s);
  CHECK_LAUNCH();
  e->have_pd = true;
  return JRR_OK;
}

extern "C" int jrr_engine_set_shape_disc(jrr_engine_t* e, const float* P, void* stream) {
  if (!e || !P) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (!(e->flags & JRR_FLAG_SHAPE_DISC)) { jrr_set_error("engine created without JRR_FLAG_SHAPE_DISC"); return JRR_ERR_STATE; }
  JRR_HIP(hipMemcpyAsync(e->Ps, P, (size_t)JRR_SHAPE_DISC_PARAMS * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  e->have_sd = true;
  return JRR_OK;
}

// =============================================================================================
// operator-level entry points
// =============================================================================================
extern "C" int jrr_rot6d_forward(const float* x, float* R, int n, void* stream) {
  if (!x || !R || n < 0) return JRR_ERR_ARG;
  if (n == 0) return JRR_OK;
  launch_rot6d_fwd(x, R, n, (hipStream_t)stream);
  CHECK_LAUNCH();
  return JRR_OK;
}
extern "C" int jrr_rot6d_backward(const float* x, const float* dR, float* dx, int n, void* stream) {
  if (!x || !dR || !dx || n < 0) return JRR_ERR_ARG;
  if (n == 0) return JRR_OK;
  launch_rot6d_bwd(x, dR, dx, n, (hipStream_t)stream);
  CHECK_LAUNCH();
  return JRR_OK;
}

/* smplx batch_rodrigues (pose2rot=True branch of the SMPL operator) */
extern "C" int jrr_rodrigues_forward(const float* aa, float* R, int n, void* stream) {
  if (!aa || !R || n < 0) return JRR_ERR_ARG;
  if (n == 0) return JRR_OK;
  launch_rodrigues_fwd(aa, R, n, (hipStream_t)stream);
  CHECK_LAUNCH();
  return JRR_OK;
}
extern "C" int jrr_rodrigues_backward(const float* aa, const float* dR, float* daa, int n, void* stream) {
  if (!aa || !dR || !daa || n < 0) return JRR_ERR_ARG;
  if (n == 0) return JRR_OK;
  launch_rodrigues_bwd(aa, dR, daa, n, (hipStream_t)stream);
  CHECK_LAUNCH();
  return JRR_OK;
}

// The per-vertex-chunk joint partials JP [nvc][3][17][BP] and skinning-adjoint partials dATp [nvcb][12][24][BP] are summed
// by their consumers (k_joints_loss, k_chain_bwd: a few pose-contiguous loads per thread); the 16 split-K slabs of
// dF^T (58 MB at 4096 poses) keep a wide reduction kernel of their own -- and so do the dA slabs when there are many of
// them (small batches: 16 slabs at 1024 poses, where k_chain_bwd has only 32 blocks to sum them with).
constexpr int MAX_SLABS_IN_CONSUMER = 8;
// conv_x6d != NULL (fused loop with the pose discriminator): the per-joint MLP adjoint shares the launch of the dF^T sum.
static void reduce_adjoint_partials(jrr_engine* e, hipStream_t s, const float* conv_x6d = nullptr, float dscale = 0.f, int nsplit = 0) {
  if (nsplit <= 0) nsplit = e->nsplit;      // (the support-tile iterations split the blend adjoint's short K range fewer ways)
  if (conv_x6d)
    launch_dconv_bwd_reduce(e->convL, conv_x6d, e->dH2T, nullptr, dscale, 1.f, e->gx, e->dsq, e->B, e->BP, e->dFTp, nsplit,
                            (size_t)KFP * e->BP, e->dF, (size_t)KFP * e->BP, s);
  else if (e->nvcb > MAX_SLABS_IN_CONSUMER) {      // both slab sums in one launch
    launch_reduce_slabs2(e->dFTp, nsplit, (size_t)KFP * e->BP, e->dF, (size_t)KFP * e->BP, e->dATp, e->nvcb, (size_t)12 * NJ * e->BP, e->dA,
                         (size_t)12 * NJ * e->BP, s);
    return;
  } else launch_reduce_slabs(e->dFTp, nsplit, (size_t)KFP * e->BP, e->dF, (size_t)KFP * e->BP, s);
  if (e->nvcb > MAX_SLABS_IN_CONSUMER)
    launch_reduce_slabs(e->dATp, e->nvcb, (size_t)12 * NJ * e->BP, e->dA, (size_t)12 * NJ * e->BP, s);
}
// k_lbs_bwd16's slabs carry joint masks instead of zero rows when k_chain_bwd sums them itself (a slab-sum launch reads every row)
static unsigned* slab_masks(jrr_engine* e) {
  return (e->m.kjs && e->m.bwd16 && e->nvcb <= MAX_SLABS_IN_CONSUMER) ? e->dmask : nullptr;
}
static void set_adjoint_slabs(jrr_engine* e, PrepBwdLaunch& L) {
  const bool pre = e->nvcb > MAX_SLABS_IN_CONSUMER;
  L.dATp = pre ? e->dA : e->dATp; L.nslabA = pre ? 1 : e->nvcb; L.strideA = (size_t)12 * NJ * e->BP; L.dFTp = e->dF;
  L.dmaskA = slab_masks(e);
}

// verts_pm: the vertices go pose-major into e->VPM (what the fused rasteriser reads) instead of the row quads of e->VTb
static int smpl_forward(jrr_engine* e, const float* x6d, const float* R, const float* betas, bool keep_vp,
                        bool keep_verts, int32_t* step_inc, hipStream_t s, const int* vmask = nullptr, const int* tl = nullptr, int ntl = 0,
                        bool verts_pm = false) {
  launch_prep_fwd(e->m, x6d, R, betas, e->FT, e->FTq, e->AT, e->R0T, e->B, e->BP, step_inc, s);
  launch_lbs_fwd(e->m, e->Jn_vi, e->FTq, e->AT, keep_vp ? e->VPb : nullptr, e->JP, verts_pm ? e->VPM : keep_verts ? e->VTb : nullptr, e->B, e->BP,
                 e->nvc, s, nullptr, tl ? nullptr : vmask, tl, ntl, verts_pm ? 1 : 0);
  if (keep_verts && !verts_pm) e->verts_partial = vmask != nullptr || tl != nullptr;
  return 0;
}

extern "C" int jrr_find_joints_forward(jrr_engine_t* e, const float* x6d, const float* R, const float* betas,
                                       float* joints, float* verts, void* stream) {
  if (!e || !betas || !joints || ((x6d == nullptr) == (R == nullptr))) { jrr_set_error("find_joints_forward: bad argument"); return JRR_ERR_ARG; }
  if (!e->have_J) { jrr_set_error("J_regressor not set"); return JRR_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  e->fwd_cached = false;
  const bool kv = (e->flags & JRR_FLAG_KEEP_VERTS) != 0;
  if (verts && !e->VTb) { jrr_set_error("return_verts needs an engine created with JRR_FLAG_KEEP_VERTS"); return JRR_ERR_STATE; }
  smpl_forward(e, x6d, R, betas, true, kv || verts, nullptr, s);
  if (verts) launch_verts_untranspose(e->VTb, verts, V * 3, V, nullptr, nullptr, e->B, e->BP, s, e->m.p2v);
  launch_joints_loss(e->JP, e->nvc, nullptr, nullptr, 0.f, joints, nullptr, nullptr, e->B, e->BP, s);
  CHECK_LAUNCH();
  return JRR_OK;
}

static int blend_adjoint_gemm(jrr_engine* e, hipStream_t s, const int* tl = nullptr, int ntl = 0, int nsplit = 0) {
  if ((e->flags & JRR_FLAG_BLEND_BF16X3) && !tl)      // side mode (include/jrr.h): split-bf16 operands, fp32 accumulation
    return launch_blend_adjoint_bf16x3(e->Dsplit, e->DVP, e->dFTp, (size_t)KFP * e->BP, e->BP, nsplit > 0 ? nsplit : e->nsplit, s);
  return launch_blend_adjoint(e->m.Dq, e->DVP, e->dFTp, (size_t)KFP * e->BP, e->BP, nsplit > 0 ? nsplit : e->nsplit, s, tl, ntl);
}
// the joint-loss iteration on the regressor's support tiles only (JRR_FLAG_SUPPORT_TILES)
static bool use_tile_list(const jrr_engine* e) {
  return (e->flags & JRR_FLAG_SUPPORT_TILES) && e->act_valid && e->have_jsup && e->jsup_fits_known && e->sil_mask == nullptr &&
         e->m.kjs && e->m.bwd16 && !(e->folded && e->fold_valid);
}

// ... per support VERTEX in one workgroup per 32-pose group (supk.h): the same condition and the support's vertex tables built
static bool use_sup_vertices(const jrr_engine* e) { return use_tile_list(e) && e->sup_valid; }

static int j_grad_from_verts(jrr_engine* e, float* dJ, hipStream_t s, float* dJs = nullptr);

extern "C" int jrr_find_joints_backward(jrr_engine_t* e, const float* x6d, const float* R, const float* betas,
                                        const float* djoints, float* dx6d, float* dR, float* dbetas, float* dJ,
                                        void* stream) {
  if (!e || !betas || !djoints || ((x6d == nullptr) == (R == nullptr))) { jrr_set_error("find_joints_backward: bad argument"); return JRR_ERR_ARG; }
  if (!e->have_J) { jrr_set_error("J_regressor not set"); return JRR_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  e->fwd_cached = false;
  launch_joints_loss(nullptr, 0, nullptr, djoints, 0.f, nullptr, nullptr, e->dJT, e->B, e->BP, s);   // (B,17,3) -> [3][18][BP]
  if (dx6d || dR || dbetas) {
    launch_lbs_bwd(e->m, e->Jn_iv, e->AT, e->VPb, e->dJT, nullptr, e->DVP, e->dATp, e->BP, e->nvcb, s, nullptr, 0, slab_masks(e));
    int rc = blend_adjoint_gemm(e, s);
    if (rc) return rc;
    reduce_adjoint_partials(e, s);
    PrepBwdLaunch L;
    L.x6d_in = x6d; L.R_in = R; L.betas_in = betas;
    set_adjoint_slabs(e, L); L.FT = e->FT; L.R0T = e->R0T; L.AT = e->AT; L.dRT = e->dRT; L.dbT = e->dbT;
    L.dx6d = dx6d; L.dR = dR; L.dbetas = dbetas;
    L.B = e->B; L.BP = e->BP;
    launch_prep_bwd(L, e->m, s);
    CHECK_LAUNCH();
  }
  if (dJ) {
    int rc = j_grad_from_verts(e, dJ, s);
    if (rc) return rc;
  }
  return JRR_OK;
}

extern "C" int jrr_smpl_vertices_backward(jrr_engine_t* e, const float* x6d, const float* R, const float* betas,
                                          const float* dverts, float* dx6d, float* dR, float* dbetas, void* stream) {
  if (!e || !betas || !dverts || ((x6d == nullptr) == (R == nullptr))) { jrr_set_error("smpl_vertices_backward: bad argument"); return JRR_ERR_ARG; }
  if (!(e->flags & JRR_FLAG_KEEP_VERTS)) { jrr_set_error("smpl_vertices_backward requires JRR_FLAG_KEEP_VERTS"); return JRR_ERR_STATE; }
  if (!e->have_J) { jrr_set_error("J_regressor not set"); return JRR_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  e->fwd_cached = false;
  // the caller's adjoint, transposed into its own [3][VP][BP] buffer (the stored vertices stay valid for a later dJ)
  launch_dverts_transpose(dverts, V * 3, e->dVTb, e->B, e->BP, s, e->m.p2v);
  launch_lbs_bwd(e->m, e->Jn_iv, e->AT, e->VPb, nullptr, e->dVTb, e->DVP, e->dATp, e->BP, e->nvcb, s, nullptr, 0, slab_masks(e));
  int rc = blend_adjoint_gemm(e, s);
  if (rc) return rc;
  reduce_adjoint_partials(e, s);
  PrepBwdLaunch L;
  L.x6d_in = x6d; L.R_in = R; L.betas_in = betas;
  set_adjoint_slabs(e, L); L.FT = e->FT; L.R0T = e->R0T; L.AT = e->AT; L.dRT = e->dRT; L.dbT = e->dbT;
  L.dx6d = dx6d; L.dR = dR; L.dbetas = dbetas;
  L.B = e->B; L.BP = e->BP;
  launch_prep_bwd(L, e->m, s);
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_smpl_posed_joints(jrr_engine_t* e, const float* betas, float* joints24, void* stream) {
  if (!e || !betas || !joints24) { jrr_set_error("smpl_posed_joints: null"); return JRR_ERR_ARG; }
  if (e->flags & JRR_FLAG_NO_MODEL) { jrr_set_error("smpl_posed_joints: engine created without a body model"); return JRR_ERR_STATE; }
  launch_posed_joints(e->m, e->AT, betas, joints24, e->B, e->BP, (hipStream_t)stream);
  CHECK_LAUNCH();
  return JRR_OK;
}

// adjoint of jrr_smpl_posed_joints through the kinematic chain (the chain forward's F^T / A^T of the most recent forward are reused)
extern "C" int jrr_smpl_posed_joints_backward(jrr_engine_t* e, const float* x6d, const float* R, const float* betas, const float* djoints24,
                                              float* dx6d, float* dR, float* dbetas, void* stream) {
  if (!e || !betas || !djoints24 || ((x6d == nullptr) == (R == nullptr))) { jrr_set_error("smpl_posed_joints_backward: bad argument"); return JRR_ERR_ARG; }
  if (e->flags & JRR_FLAG_NO_MODEL) { jrr_set_error("smpl_posed_joints_backward: engine created without a body model"); return JRR_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  e->fwd_cached = false;
  launch_posed_joints_bwd(e->m, e->AT, betas, djoints24, e->dA, e->dbT, e->B, e->BP, s);
  JRR_HIP(hipMemsetAsync(e->dF, 0, (size_t)KFP * e->BP * sizeof(float), s));      // the posed joints do not read the blend features
  PrepBwdLaunch L;
  L.x6d_in = x6d; L.R_in = R; L.betas_in = betas;
  L.dATp = e->dA; L.nslabA = 1; L.strideA = (size_t)12 * NJ * e->BP; L.dFTp = e->dF; L.dmaskA = nullptr;
  L.FT = e->FT; L.R0T = e->R0T; L.AT = e->AT; L.dRT = e->dRT; L.dbT = e->dbT;
  L.gb_extra = e->dbT;                                                             // (B,10): the direct term through J_j(beta)
  L.dx6d = dx6d; L.dR = dR; L.dbetas = dbetas;
  L.B = e->B; L.BP = e->BP;
  launch_prep_bwd(L, e->m, s);
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_joint_loss(const float* joints, const float* gt_mm, float weight, int batch, int batch_norm,
                              float* sqerr, float* djoints, void* stream) {
  if (!joints || !gt_mm || batch <= 0 || batch_norm <= 0) return JRR_ERR_ARG;
  const float scale = (float)(2.0 * (double)weight / ((double)batch_norm * 51.0));
  launch_joint_loss_plain(joints, gt_mm, scale, sqerr, djoints, batch, (hipStream_t)stream);
  CHECK_LAUNCH();
  return JRR_OK;
}

// ---- pose discriminator --------------------------------------------------------------------
// Six launches per forward + input gradient (scripts/discriminator.py:32-54 and its adjoint):
//   k_dconv_fwd (per-joint MLP, MFMA)  ->  fc0 GEMM (+bias, ReLU)  ->  fc2 GEMM (+bias, ReLU, the fc4 dot product
//   w4 . a2 as per-column partials in the epilogue; what it stores is relu'(a2) as 0 / 1: nothing else of a2 is read again)
//   ->  fc2 adjoint GEMM on that indicator (dz from the partial dots in the prologue multiplies the finished column sums
//   in the epilogue; fc2.w pre-scaled by w4: the rank-one output-layer adjoint is never materialised)  ->  fc0 adjoint
//   GEMM  ->  k_dconv_bwd.
// All four GEMMs: exact 128x64 tiles (512 workgroups at 4096 poses), 3-deep LDS-DMA ring.
// The loop path keeps every activation in quads [row/4][pose][4] (k_disc_gemm); the weight-gradient path of the outer
// step (disc_backward_params) needs row-major activations for its transposes / row sums and runs the row-major kernels.
// conv_done: the per-joint MLP already ran (fused into the chain-forward launch, launch_prep_fwd_dconv)
static int disc_forward(jrr_engine* e, const float* x6d, float* out, hipStream_t s, bool quad = true, bool conv_done = false) {
  if (!conv_done) launch_disc_conv_fwd(e->convL, x6d, e->H2T, out, e->B, e->BP, s, quad ? 1 : 0);
  GemmArgs g;
  g.mask = nullptr; g.split_stride = 0; g.N = e->BP; g.ldb = e->BP; g.ldo = e->BP;
  g.A = quad ? e->W0Tq : e->W0T; g.lda = 1024; g.Bm = e->H2T; g.Out = e->A1T; g.bias = e->Pd + DP_FC0_B; g.M = 1024; g.K = 768;
  int rc = quad ? launch_disc_gemm_q(g, EPI_BIAS_RELU, 0, s) : launch_gemm_128x64(g, EPI_BIAS_RELU, 1, s);
  if (rc) return rc;
  g.A = quad ? e->W2Tq : e->W2T; g.lda = 1024; g.Bm = e->A1T; g.Out = e->A2T; g.bias = e->Pd + DP_FC2_B; g.M = 1024; g.K = 1024;
  if (!quad) return launch_gemm_128x64(g, EPI_BIAS_RELU, 1, s);
  g.dotw = e->Pd + DP_FC4_W; g.dot_out = e->zpart;
  return launch_disc_gemm_q(g, EPI_BIAS_RELU_DOT, 0, s);
}

// skip_conv: the caller runs the per-joint MLP adjoint itself (fused with the dF^T slab sum, launch_dconv_bwd_reduce)
static int disc_backward_input(jrr_engine* e, const float* x6d, float* out, const float* gout, float scale,
                               float target, float* gx, hipStream_t s, float* sq = nullptr, bool skip_conv = false) {
  GemmArgs g;
  g.bias = nullptr; g.split_stride = 0; g.N = e->BP; g.ldb = e->BP; g.ldo = e->BP;
  // dA1T[k][b] = relu'(A1T) * sum_n (fc4.w[n] fc2.w[n][k]) relu'(A2T[n][b]) dz[b]
  g.A = e->W2sq; g.lda = 1024; g.Bm = e->A2T; g.Out = e->dA1T; g.mask = e->A1T; g.M = 1024; g.K = 1024;
  g.zpart = e->zpart; g.nzpart = 16; g.zbias = e->Pd + DP_FC4_B; g.gout = gout; g.gout_ld = 25; g.scale = scale; g.target = target;
  g.nvalid = e->B; g.sq0 = sq; g.out0 = out; g.out0_ld = 25;
  int rc = launch_disc_gemm_q(g, EPI_MASK, 2, s);
  if (rc) return rc;
  // dH2T[k][b] = sum_n fc0.w[n][k] dA1T[n][b]
  GemmArgs h;
  h.bias = nullptr; h.split_stride = 0; h.N = e->BP; h.ldb = e->BP; h.ldo = e->BP;
  h.A = e->W0q; h.lda = 768; h.Bm = e->dA1T; h.Out = e->dH2T; h.mask = nullptr; h.M = 768; h.K = 1024;
  rc = launch_disc_gemm_q(h, EPI_STORE, 0, s);
  if (rc) return rc;
  if (!skip_conv) launch_disc_conv_bwd(e->convL, x6d, e->dH2T, gout, scale, target, gx, e->B, e->BP, s, sq, 1);
  return 0;
}

extern "C" int jrr_pose_disc_forward(jrr_engine_t* e, const float* x6d, float* out, void* stream) {
  if (!e || !x6d || !out) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (!e->have_pd) { jrr_set_error("pose discriminator not set"); return JRR_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  int rc = disc_forward(e, x6d, out, s);
  if (rc) return rc;
  launch_disc_z_finish(e->zpart, 16, e->BP, e->Pd + DP_FC4_B, out, e->B, s);
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_pose_disc_backward_input(jrr_engine_t* e, const float* x6d, float weight, float target, float* dx,
                                            void* stream) {
  if (!e || !x6d || !dx) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (!e->have_pd) { jrr_set_error("pose discriminator not set"); return JRR_ERR_STATE; }
  const float scale = (float)(2.0 * (double)weight / ((double)e->bnorm * 25.0));
  int rc = disc_backward_input(e, x6d, nullptr, nullptr, scale, target, dx, (hipStream_t)stream);
  if (rc) return rc;
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_pose_disc_vjp_input(jrr_engine_t* e, const float* x6d, const float* gout, float* dx, void* stream) {
  if (!e || !x6d || !gout || !dx) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (!e->have_pd) { jrr_set_error("pose discriminator not set"); return JRR_ERR_STATE; }
  int rc = disc_backward_input(e, x6d, nullptr, gout, 0.f, 0.f, dx, (hipStream_t)stream);
  if (rc) return rc;
  CHECK_LAUNCH();
  return JRR_OK;
}

// weight gradients of the pose discriminator: either of the MSE against `target` (gout == NULL; scale = 2/(bnorm*25))
// or the vector-Jacobian product for an arbitrary upstream gradient gout (B,25)
static int disc_backward_params(jrr_engine* e, const float* x6d, const float* gout, float scale, float target, float* dP,
                                float* sqerr, hipStream_t s) {
  int rc = disc_forward(e, x6d, e->dsc, s, false);        // row-major activations for the transposes / row sums below
  if (rc) return rc;
  launch_disc_out(e->Pd, e->A2T, e->dsc, e->dA2T, gout, scale, target, e->B, e->BP, s, e->dz0);
  if (sqerr) launch_sqerr_rows(e->dsc, 25, target, sqerr, e->B, s);
  // fc4: dw[n] += sum_b A2T[n][b] dz0[b] ; db += sum_b dz0[b]
  launch_rowdot_accum(e->A2T, e->BP, e->dz0, dP + DP_FC4_W, 1024, e->BP, s);
  launch_rowdot_accum(e->dz0, e->BP, nullptr, dP + DP_FC4_B, 1, e->BP, s);
  // fc2: dW[n][k] += sum_b dA2T[n][b] A1T[k][b] (pose-major copies feed the K-major GEMM) ; db[n] += sum_b dA2T[n][b]
  launch_transpose(e->dA2T, e->TrA, 1024, e->BP, s);
  launch_transpose(e->A1T, e->TrB, 1024, e->BP, s);
  GemmArgs g;
  g.bias = nullptr; g.mask = nullptr; g.split_stride = 0;
  // the pose dimension is the reduction: split it so that the 64 output tiles become >= 512 workgroups; partial
  // slabs, then a wide accumulate-reduce into the flat gradient (deterministic, no atomics)
  const int wsplit = e->BP >= 4096 ? 8 : e->BP >= 1024 ? 4 : e->BP >= 256 ? 2 : 1;
  g.A = e->TrA; g.lda = 1024; g.Bm = e->TrB; g.ldb = 1024; g.Out = e->wgs; g.ldo = 1024; g.M = 1024; g.N = 1024; g.K = e->BP;
  g.split_stride = (size_t)1024 * 1024;
  rc = launch_gemm_128(g, EPI_STORE, wsplit, s);
  if (rc) return rc;
  launch_reduce_slabs(e->wgs, wsplit, (size_t)1024 * 1024, dP + DP_FC2_W, (size_t)1024 * 1024, s, 1);
  g.split_stride = 0;
  launch_rowdot_accum(e->dA2T, e->BP, nullptr, dP + DP_FC2_B, 1024, e->BP, s);
  // back through fc2
  g.A = e->Pd + DP_FC2_W; g.lda = 1024; g.Bm = e->dA2T; g.ldb = e->BP; g.Out = e->dA1T; g.ldo = e->BP; g.mask = e->A1T;
  g.M = 1024; g.N = e->BP; g.K = 1024;
  rc = launch_gemm_128x64(g, EPI_MASK, 1, s);
  if (rc) return rc;
  // fc0
  launch_transpose(e->dA1T, e->TrA, 1024, e->BP, s);
  launch_transpose(e->H2T, e->TrB, 768, e->BP, s);
  g.mask = nullptr;
  g.A = e->TrA; g.lda = 1024; g.Bm = e->TrB; g.ldb = 768; g.Out = e->wgs; g.ldo = 768; g.M = 1024; g.N = 768; g.K = e->BP;
  g.split_stride = (size_t)1024 * 768;
  rc = launch_gemm_128(g, EPI_STORE, wsplit, s);
  if (rc) return rc;
  launch_reduce_slabs(e->wgs, wsplit, (size_t)1024 * 768, dP + DP_FC0_W, (size_t)1024 * 768, s, 1);
  g.split_stride = 0;
  launch_rowdot_accum(e->dA1T, e->BP, nullptr, dP + DP_FC0_B, 1024, e->BP, s);
  g.A = e->Pd + DP_FC0_W; g.lda = 768; g.Bm = e->dA1T; g.ldb = e->BP; g.Out = e->dH2T; g.ldo = e->BP; g.M = 768; g.N = e->BP; g.K = 1024;
  rc = launch_gemm_128x64(g, EPI_STORE, 1, s);
  if (rc) return rc;
  {   // conv / head weight gradients: one slab per wave [pose group][joint][1280], reduced in two wide steps
      // (over the pose groups, then over the joints) into the flat gradient
    const int ng = e->BP / 64;
    float* slab_shared = e->wgs;
    float* slab_heads = slab_shared + (size_t)NJ * ng * 1280;
    float* tmp = slab_heads + (size_t)ng * 792;
    launch_disc_conv_bwd_params(e->Pd, x6d, e->dH2T, gout, scale, target, slab_shared, slab_heads, e->B, e->BP, s);
    launch_reduce_slabs(slab_shared, ng, (size_t)NJ * 1280, tmp, (size_t)NJ * 1280, s, 0);
    launch_reduce_slabs(tmp, NJ, 1280, dP + DP_CONV0_W, 1280, s, 1);
    launch_reduce_slabs(slab_heads, ng, 792, dP + DP_HEADS, 792, s, 1);
  }
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_pose_disc_backward_params(jrr_engine_t* e, const float* x6d, float target, float* dP, float* sqerr,
                                             void* stream) {
  if (!e || !x6d || !dP) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (!e->have_pd) { jrr_set_error("pose discriminator not set"); return JRR_ERR_STATE; }
  return disc_backward_params(e, x6d, nullptr, (float)(2.0 / ((double)e->bnorm * 25.0)), target, dP, sqerr, (hipStream_t)stream);
}

extern "C" int jrr_pose_disc_vjp_params(jrr_engine_t* e, const float* x6d, const float* gout, float* dP, void* stream) {
  if (!e || !x6d || !gout || !dP) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (!e->have_pd) { jrr_set_error("pose discriminator not set"); return JRR_ERR_STATE; }
  return disc_backward_params(e, x6d, gout, 0.f, 0.f, dP, nullptr, (hipStream_t)stream);
}

extern "C" int jrr_shape_disc_vjp_params(jrr_engine_t* e, const float* betas, const float* gout, float* dP, void* stream) {
  if (!e || !betas || !gout || !dP) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (!e->have_sd) { jrr_set_error("shape discriminator not set"); return JRR_ERR_STATE; }
  launch_shape_disc_bwd_params(e->Ps, betas, gout, 0.f, 0.f, dP, nullptr, e->B, (hipStream_t)stream);
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_shape_disc_backward_params(jrr_engine_t* e, const float* betas, float target, float* dP, float* sqerr,
                                              void* stream) {
  if (!e || !betas || !dP) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (!e->have_sd) { jrr_set_error("shape discriminator not set"); return JRR_ERR_STATE; }
  const float scale = (float)(2.0 / ((double)e->bnorm * 1.0));
  launch_shape_disc_bwd_params(e->Ps, betas, nullptr, scale, target, dP, sqerr, e->B, (hipStream_t)stream);
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_shape_disc_forward(jrr_engine_t* e, const float* betas, float* out, void* stream) {
  if (!e || !betas || !out) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (!e->have_sd) { jrr_set_error("shape discriminator not set"); return JRR_ERR_STATE; }
  launch_shape_disc(e->Ps, betas, out, nullptr, 0.f, 0.f, e->B, (hipStream_t)stream);
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_shape_disc_vjp_input(jrr_engine_t* e, const float* betas, const float* gout, float* dbetas, void* stream) {
  if (!e || !betas || !gout || !dbetas) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (!e->have_sd) { jrr_set_error("shape discriminator not set"); return JRR_ERR_STATE; }
  launch_shape_disc(e->Ps, betas, nullptr, dbetas, 0.f, 0.f, e->B, (hipStream_t)stream, gout);
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_refine_aux_losses(jrr_engine_t* e, float* pose_disc_sq, float* shape_disc_sq, void* stream) {
  if (!e) return JRR_ERR_ARG;
  e->fwd_cached = false;
  hipStream_t s = (hipStream_t)stream;
  if (pose_disc_sq) {
    if (!((e->flags & JRR_FLAG_POSE_DISC) && e->have_pd)) { jrr_set_error("pose discriminator term not active"); return JRR_ERR_STATE; }
    launch_colsum(e->dsq, 25, e->BP, pose_disc_sq, e->B, s);
  }
  if (shape_disc_sq) {
    if (!((e->flags & JRR_FLAG_SHAPE_DISC) && e->have_sd)) { jrr_set_error("shape discriminator term not active"); return JRR_ERR_STATE; }
    JRR_HIP(hipMemcpyAsync(shape_disc_sq, e->ssq, (size_t)e->B * 4, hipMemcpyDeviceToDevice, s));
  }
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_adam_step(float* p, const float* g, float* m, float* v, size_t n, const int32_t* step, float lr,
                             float beta1, float beta2, float eps, void* stream) {
  if (!p || !g || !m || !v || !step) return JRR_ERR_ARG;
  if (n == 0) return JRR_OK;
  launch_adam_flat(p, g, m, v, n, step, lr, beta1, beta2, eps, (hipStream_t)stream);
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_evaluate(const float* pred, const float* target_mm, float* err, float* err_pa, int batch, void* stream) {
  if (!pred || !target_mm || !err || !err_pa || batch <= 0) return JRR_ERR_ARG;
  launch_evaluate(pred, target_mm, err, err_pa, batch, (hipStream_t)stream);
  CHECK_LAUNCH();
  return JRR_OK;
}

// =============================================================================================
// 2-D reprojection (row f1)
// =============================================================================================
extern "C" int jrr_project_joints(const float* joints, const float* cam, float* j2d, int batch, void* stream) {
  if (!joints || !cam || !j2d || batch <= 0) return JRR_ERR_ARG;
  launch_project_joints(joints, cam, j2d, batch, (hipStream_t)stream);
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_engine_set_reprojection(jrr_engine_t* e, const float* gt_j2d, float* cam, float* cam_m, float* cam_v) {
  if (!e) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (gt_j2d && (!cam || !cam_m || !cam_v)) { jrr_set_error("set_reprojection: cam / cam_m / cam_v required"); return JRR_ERR_ARG; }
  e->gt_j2d = gt_j2d; e->cam = cam; e->cam_m = cam_m; e->cam_v = cam_v;
  return JRR_OK;
}

extern "C" int jrr_camera_prefit(jrr_engine_t* e, const float* x6d, const float* betas, const float* gt_j2d, float* cam,
                                 int n_steps, float lr, float* sq2d, void* stream) {
  if (!e || !x6d || !betas || !gt_j2d || !cam || n_steps < 0) return JRR_ERR_ARG;
  if (!e->have_J) { jrr_set_error("J_regressor not set"); return JRR_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  e->fwd_cached = false;
  smpl_forward(e, x6d, nullptr, betas, false, false, nullptr, s);
  launch_joints_loss(e->JP, e->nvc, nullptr, nullptr, 0.f, e->joints, nullptr, nullptr, e->B, e->BP, s);
  const float scale2d = (float)(2.0 / ((double)e->bnorm * 34.0));     // optimize.py:193 unweighted MSE
  launch_camera_fit(e->joints, gt_j2d, cam, scale2d, n_steps, lr, sq2d, e->B, s);
  CHECK_LAUNCH();
  return JRR_OK;
}

// =============================================================================================
// soft silhouette (row f2)
// =============================================================================================
static int sil_check(jrr_engine* e) {
  if (!(e->flags & JRR_FLAG_SILHOUETTE)) { jrr_set_error("engine created without JRR_FLAG_SILHOUETTE"); return JRR_ERR_STATE; }
  if (!e->m.faces) { jrr_set_error("model has no faces (jrr_model_set_faces)"); return JRR_ERR_STATE; }
  return 0;
}

extern "C" int jrr_silhouette_forward(jrr_engine_t* e, const float* verts, const float* cam, float* alpha, void* stream) {
  if (!e || !verts || !cam || !alpha) return JRR_ERR_ARG;
  int rc = sil_check(e);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  e->fwd_cached = false;
  launch_sil_project(verts, V * 3, cam, e->ndc, e->B, s, e->sil);
  launch_sil_raster(e->ndc, e->m.faces_pk, e->m.nfaces, e->cover, e->ncover, alpha, e->B, s, e->sil);
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_silhouette_backward(jrr_engine_t* e, const float* galpha, float* dverts, float* dcam, void* stream) {
  if (!e || !galpha || !dverts || !dcam) return JRR_ERR_ARG;
  int rc = sil_check(e);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  e->fwd_cached = false;
  launch_sil_bwd(e->ndc, e->m.faces, e->cover, e->ncover, nullptr, galpha, 0.f, dverts, V * 3, dcam, 0, e->B, s, e->sil);
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_silhouette_pix_to_face(jrr_engine_t* e, int32_t* p2f, void* stream) {
  if (!e || !p2f) return JRR_ERR_ARG;
  int rc = sil_check(e);
  if (rc) return rc;
  launch_sil_pix_to_face(e->cover, e->ncover, p2f, e->B, (hipStream_t)stream, e->sil);
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_engine_set_silhouette(jrr_engine_t* e, const float* mask, float* cam, float* cam_m, float* cam_v) {
  if (!e) return JRR_ERR_ARG;
  e->fwd_cached = false;
  if (mask) {
    int rc = sil_check(e);
    if (rc) return rc;
    if (!cam || !cam_m || !cam_v) { jrr_set_error("set_silhouette: cam / cam_m / cam_v required"); return JRR_ERR_ARG; }
    if (e->sil != 224 && e->sil != 256) {
      jrr_set_error("set_silhouette: the silhouette term inside the loop is built for 224 x 224 and 256 x 256 images (this engine: %d); "
                    "the other sizes serve the stand-alone renderer", e->sil);
      return JRR_ERR_STATE;
    }
    e->cam = cam; e->cam_m = cam_m; e->cam_v = cam_v;
  }
  e->sil_mask = mask;
  e->smask_valid = false;          // sum(mask^2) per pose: recomputed on the stream of the next jrr_refine_run
  return JRR_OK;
}

// The silhouette term exactly as the fused inner loop evaluates it (k_sil_raster<true>: in-kernel projection from the
// row-quad vertex buffer, packed fixed-point adjoint, write-back over the vertex pieces), as an operator: SMPL forward of
// (x6d, betas), then loss and gradient w.r.t. the vertices and the camera.
extern "C" int jrr_silhouette_loss_grad(jrr_engine_t* e, const float* x6d, const float* betas, const float* cam,
                                        const float* mask, float* sqsil, float* dverts, float* dcam, void* stream) {
  if (!e || !x6d || !betas || !cam || !mask) { jrr_set_error("silhouette_loss_grad: null"); return JRR_ERR_ARG; }
  int rc = sil_check(e);
  if (rc) return rc;
  if (!e->have_J) { jrr_set_error("J_regressor not set"); return JRR_ERR_STATE; }
  if (!e->VTb) { jrr_set_error("silhouette_loss_grad needs JRR_FLAG_KEEP_VERTS"); return JRR_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  e->fwd_cached = false;
  smpl_forward(e, x6d, nullptr, betas, true, true, nullptr, s, nullptr, nullptr, 0, true);      // vertices pose-major, as in the loop
  launch_mask_sq(mask, e->smask, e->B, s, e->sil);
  e->smask_valid = false;
  const float silscale = (float)(2.0 * 100.0 / ((double)e->bnorm * (double)e->sil * (double)e->sil));      // optimize.py:252 weight 100
  { int rcs = launch_sil_raster_adj(e->VTb, e->BP, cam, e->m.faces_int_pk ? e->m.faces_int_pk : e->m.faces_pk, e->m.nfaces, mask, e->smask, e->cover,
                                    e->ncover, e->sqsil, silscale, e->gcam, 0, e->B, s, e->sil, e->VPM);
    if (rcs) return rcs; }      // (sizes other than 224 / 256: the stand-alone forward / backward pair only)
  if (sqsil) JRR_HIP(hipMemcpyAsync(sqsil, e->sqsil, (size_t)e->B * 4, hipMemcpyDeviceToDevice, s));
  if (dverts) launch_verts_untranspose(e->VTb, dverts, V * 3, V, nullptr, nullptr, e->B, e->BP, s, e->m.p2v);
  if (dcam) JRR_HIP(hipMemcpyAsync(dcam, e->gcam, (size_t)e->B * 3 * 4, hipMemcpyDeviceToDevice, s));
  CHECK_LAUNCH();
  return JRR_OK;
}

// joints^T partials from the STORED vertices: JPv[split][r][32][BP] = sum_{v in split} Jn[i,v] verts_r[v,b]
// (both operands in vertex quads: Jn_q [VP/4][32][4], VTb [3][VP/4][BP][4]; rows i >= 17 of Jn_q are zero)
static int joints_from_stored_verts(jrr_engine* e, hipStream_t s, int32_t* step_inc = nullptr) {
  // slab layout [split][plane][32][BP], what k_joints_loss reads with jp_rows = 32.  The support-restricted kernel writes ONE
  // complete slab (slab 0) when the regressor's support lists fit (device flag jsup.flag, which k_joints_loss also reads to
  // sum one slab only); otherwise it returns at once and the dense product below does the work -- and vice versa.
  if (e->have_jsup) launch_rejoints_sparse(e->jsup, e->VTb, e->dFTp, e->BP, s, step_inc);
  // the host KNOWS that the lists fit (jrr_j_support_info; J steps only shrink the support): the dense product need not even be
  // enqueued (an idle launch still costs ~4.7 us of stream time)
  if (e->have_jsup && e->jsup_fits_known) return 0;
  return launch_gemm_q32(e->Jn_q, 32, 0, e->VTb, e->BP, (size_t)VP * e->BP, e->dFTp, e->BP, (size_t)3 * 32 * e->BP,
                         (size_t)32 * e->BP, e->BP, VP, 3, e->nsplit, s, e->have_jsup ? e->jsup.flag : nullptr);
}

// =============================================================================================
// fused inner loop (scripts/optimize.py:220-265)
// =============================================================================================
// k-th record of the loss history: the five weighted terms of scripts/optimize.py:252-253 as this rank's share of the
// global means (sum over the local poses / the global denominators), from the per-pose sums the iteration left behind
__global__ void __launch_bounds__(1024) k_loss_record(const float* __restrict__ sqj, const float* __restrict__ sq2d,
                                                      const float* __restrict__ sqsil, const float* __restrict__ dsq,
                                                      const float* __restrict__ ssq, int B, int BP, float bnorm,
                                                      float* __restrict__ rec, float npix) {
  __shared__ float red[5][1024];
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < B; b += 1024) {
    if (sq2d) acc[0] += sq2d[b];
    if (sqsil) acc[1] += sqsil[b];
    acc[2] += sqj[b];
    if (dsq) { float a = 0.f; for (int k = 0; k < 25; ++k) a += dsq[(size_t)k * BP + b]; acc[3] += a; }
    if (ssq) acc[4] += ssq[b];
  }
  for (int t = 0; t < 5; ++t) red[t][threadIdx.x] = acc[t];
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) for (int t = 0; t < 5; ++t) red[t][threadIdx.x] += red[t][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    rec[0] = red[0][0] / (bnorm * 34.f) * 0.01f;            // loss_j2d / 100
    rec[1] = red[1][0] / (bnorm * npix) * 100.f;            // silhouette_loss * 100
    rec[2] = red[2][0] / (bnorm * 51.f) * 10000.f;          // joint_loss * 10000
    rec[3] = red[3][0] / (bnorm * 25.f) * 10.f;             // pose_discriminated_loss * 10
    rec[4] = red[4][0] / bnorm * 10.f;                      // shape_discriminated_loss * 10
  }
}

extern "C" int jrr_engine_set_loss_history(jrr_engine_t* e, float* hist_dev, int capacity_records, int every) {
  if (!e || (hist_dev && (capacity_records <= 0 || every <= 0))) return JRR_ERR_ARG;
  e->hist = hist_dev; e->hist_cap = hist_dev ? capacity_records : 0; e->hist_every = every; e->hist_n = 0; e->hist_iter = 0;
  return JRR_OK;
}
extern "C" int jrr_engine_loss_history_count(const jrr_engine_t* e) { return e ? e->hist_n : JRR_ERR_ARG; }

static int j_step_local(jrr_engine* e, const float* x6d, const float* betas, const float* gt_mm, float* dJ, float* sqerr, hipStream_t s,
                        float* joints = nullptr, float* dJs = nullptr, bool support_verts = false);
static int j_step_apply(jrr_engine* e, float* J, const float* dJ, float* m, float* v, int32_t* step, float lr, const float* mask,
                        hipStream_t s, const float* dJs = nullptr);

// Shared-parameter state of the in-call J steps (jrr_refine_run_j_steps)
struct JStepArgs { int every; float* J; float* m; float* v; int32_t* step; float lr; const float* mask; float* sqerr; bool reuse; };

static int refine_run_impl(jrr_engine_t* e, float* x6d, float* betas, const float* gt_mm, float* adam_m,
                           float* adam_v, int32_t* step, float lr, int n_iters, float* sqerr, bool reuse_first,
                           const JStepArgs* js, void* stream) {
  if (!e || !x6d || !betas || !gt_mm || !adam_m || !adam_v || !step || n_iters < 0) { jrr_set_error("refine_run: bad argument"); return JRR_ERR_ARG; }
  if (!e->have_J) { jrr_set_error("J_regressor not set"); return JRR_ERR_STATE; }
  const bool pd = (e->flags & JRR_FLAG_POSE_DISC) && e->have_pd;
  const bool sd = (e->flags & JRR_FLAG_SHAPE_DISC) && e->have_sd;
  hipStream_t s = (hipStream_t)stream;
  const float jscale = (float)(2.0 * 10000.0 / ((double)e->bnorm * 51.0));   // optimize.py:252 weight 10000
  const float dscale = (float)(2.0 * 10.0 / ((double)e->bnorm * 25.0));      // optimize.py:253 weight 10
  const float sscale = (float)(2.0 * 10.0 / ((double)e->bnorm * 1.0));
  if (reuse_first) {
    // the caller states that the previous call on this engine was the J step on exactly these poses and that nothing
    // has written them since; everything the engine can check is checked
    const bool ok = e->fwd_cached && e->fc_x6d == x6d && e->fc_betas == betas && e->VTb != nullptr;
    if (!ok) {
      jrr_set_error("refine_run_after_j_step: the previous call on this engine was not jrr_j_regressor_grad on the same pose buffers");
      return JRR_ERR_STATE;
    }
  }
  bool reuse_next = reuse_first;
  bool pre_done = false;       // k_tail_step of the previous iteration ran this iteration's chain forward [+ MLP forward] and counted its step
  bool h2t_ready = false;      // k_sup_step of the previous iteration left the per-joint MLP forward of the current poses in H2T
  for (int it = 0; it < n_iters; ++it) {
    const bool folded = e->folded && e->fold_valid;
    const bool listed = use_tile_list(e);
    const int* tl = listed ? e->act_list : nullptr;
    const int ntl = listed ? e->nact : 0;
    // split-K of the blend adjoint: its K range is 6 chunks per listed tile -- at least JRR_ADJ_CHUNKS (6) chunks per split, so that a
    // handful of tiles does not leave 16 slabs of 3.7 MB (at 4096 poses) for the slab sum to read (6 tiles, 4096 poses: 6 splits
    // 0.3575 ms per iteration, 16 splits 0.3623, 4 splits 0.3618, 2 splits 0.377)
    static const int adj_chunks = [] { const char* v = getenv("JRR_ADJ_CHUNKS"); return v ? std::max(1, atoi(v)) : 6; }();
    // (and never fewer than ~128 workgroups in the launch: small batches have few 128-pose tiles)
    const int ns_adj = listed ? std::max(1, std::min(e->nsplit, std::max((6 * ntl + adj_chunks - 1) / adj_chunks, (128 * 128 + e->BP - 1) / e->BP))) : 0;
    // The J step that preceded this call ran the SMPL forward on exactly these poses (jrr_j_regressor_grad keeps
    // v_posed, the skinning transforms and the vertices): the first iteration re-regresses the joints with the NEW
    // regressor from the stored vertices instead of repeating the 0.47 ms forward (jrr_refine_run_after_j_step).
    // the support-vertex iteration (supk.h): forward, loss and backward of a 32-pose group in one workgroup; its forward costs less than
    // re-regressing the stored vertices, so a pending reuse is simply not taken
    const bool supv = use_sup_vertices(e);
    // JRR_SUPPORT_FUSED=2 (verification / A-B knob): the same iteration as separate launches (chain forward, k_sup_iter, per-joint MLP
    // adjoint, chain adjoint) instead of the composed kernel k_sup_step
    static const bool sup_split = [] { const char* v = getenv("JRR_SUPPORT_FUSED"); return v && v[0] == '2'; }();
    if (supv && !sup_split) {
      // ---- ONE launch per iteration and pose group (+ the four discriminator GEMMs before it): prep.hip k_sup_step ----
      e->fwd_cached = false; reuse_next = false;
      prof_mark(e, 0, s);
      // per-joint MLP forward: left behind by the previous iteration's launch, except before the first one of a call
      if (pd && !h2t_ready) launch_disc_conv_fwd(e->convL, x6d, e->H2T, nullptr, e->B, e->BP, s, 1);
      prof_mark(e, 0, s);
      // Small batches with the pose discriminator: the iteration's GEMM-independent half (k_sup_step<1>: chain forward, support-vertex
      // forward / loss / backward -- 38 of the launch's 70 us, on B / 32 workgroups) runs on the engine's side stream BESIDE the four GEMM
      // launches, the rest (k_sup_step<2>) behind both.  Fork: the side stream waits for everything enqueued on `s` so far (the previous
      // iteration's pose update); join: `s` waits for the side launch before the second half.  Same bits as the composed kernel.  Same
      // box, ms per iteration composed -> overlapped: 256 poses 0.134 -> 0.112, 512: 0.135 -> 0.126, 1024: 0.162 -> 0.160, 4096:
      // 0.336 -> 0.345 (the half-chip launch takes CUs from GEMMs that fill the chip): on up to 512 poses.  JRR_SUP_OVERLAP=0 / 1: never / always.
      static const int sup_overlap_env = [] { const char* v = getenv("JRR_SUP_OVERLAP"); return v ? (v[0] == '1' ? 1 : 0) : -1; }();
      const bool overlap = pd && (sup_overlap_env < 0 ? e->B <= 512 : sup_overlap_env == 1);
      if (overlap && !e->side) {
        if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess) { e->side = nullptr; jrr_set_error("side stream: %s", hipGetErrorString(hipGetLastError())); return JRR_ERR_HIP; }
        JRR_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        JRR_HIP(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
      }
      auto fill = [&](SupStepLaunch& q, PrepBwdLaunch& L) {
        q.t = e->sup; q.nsv = e->sup_nsv; q.Jn_vi = e->Jn_vi; q.gt_mm = gt_mm; q.scale = jscale;
        q.FT = e->FT; q.FTq = e->FTq; q.AT = e->AT; q.R0T = e->R0T; q.joints_out = e->joints; q.sqerr = sqerr ? sqerr : e->sqerr;
        q.dA = e->dA; q.dF = e->dF;
        if (pd) {
          q.conv_img = e->convL; q.dH2T = e->dH2T; q.dscale = dscale; q.gx = e->gx; q.dsq = e->dsq;
          q.H2T_next = (it + 1 < n_iters) ? e->H2T : nullptr;
        }
        q.step = step; q.arrive = e->step_scratch;
        L.x6d_in = x6d; L.betas_in = betas; L.gx_extra = pd ? e->gx : nullptr; L.gb_extra = sd ? e->gb : nullptr;
        L.x6d_io = x6d; L.betas_io = betas; L.adam_m = adam_m; L.adam_v = adam_v; L.step = step; L.lr = lr; L.B = e->B; L.BP = e->BP;
        if (e->gt_j2d) {      // 2-D term, weight 1/100 (optimize.py:231-233,252): the camera translation is a parameter of the same Adam
          q.gt_j2d = e->gt_j2d; q.cam = e->cam; q.gcam = e->gcam; q.sq2d = e->sq2d; q.scale2d = (float)(2.0 * 0.01 / ((double)e->bnorm * 34.0));
          L.gcam = e->gcam; L.cam_io = e->cam; L.cam_m = e->cam_m; L.cam_v = e->cam_v;
        }
      };
      if (overlap) {
        SupStepLaunch q1; PrepBwdLaunch L1;
        fill(q1, L1);
        JRR_HIP(hipEventRecord(e->ev_fork, s));
        JRR_HIP(hipStreamWaitEvent(e->side, e->ev_fork, 0));
        int rc1 = launch_sup_step(e->m, q1, L1, e->side, 1);
        if (rc1) return rc1;
        JRR_HIP(hipEventRecord(e->ev_join, e->side));
      }
      if (pd) {
        prof_mark(e, 5, s);
        int rcd = disc_forward(e, x6d, nullptr, s, true, true);
        if (rcd) return rcd;
        rcd = disc_backward_input(e, x6d, nullptr, nullptr, dscale, 1.f, e->gx, s, e->dsq, true);
        prof_mark(e, 5, s);
        if (rcd) return rcd;
      }
      if (sd) {
        prof_mark(e, 6, s);
        launch_shape_disc(e->Ps, betas, nullptr, e->gb, sscale, 1.f, e->B, s, nullptr, e->ssq);
        prof_mark(e, 6, s);
      }
      if (overlap) JRR_HIP(hipStreamWaitEvent(s, e->ev_join, 0));
      prof_mark(e, 1, s);
      SupStepLaunch q;
      q.t = e->sup; q.nsv = e->sup_nsv; q.Jn_vi = e->Jn_vi; q.gt_mm = gt_mm; q.scale = jscale;
      const bool js_next = js && (it + 1) % js->every == 0;
      PrepBwdLaunch L;
      fill(q, L);
      int rcq = launch_sup_step(e->m, q, L, s, overlap ? 2 : 0);
      if (rcq) return rcq;
      h2t_ready = pd && q.H2T_next != nullptr;
      prof_mark(e, 1, s);
      if (e->hist) {
        if (e->hist_iter % e->hist_every == 0 && e->hist_n < e->hist_cap) {
          hipLaunchKernelGGL(k_loss_record, dim3(1), dim3(1024), 0, s, sqerr ? sqerr : e->sqerr, e->gt_j2d ? e->sq2d : nullptr, nullptr, pd ? e->dsq : nullptr,
                             sd ? e->ssq : nullptr, e->B, e->BP, (float)e->bnorm, e->hist + (size_t)e->hist_n * 5, (float)(e->sil * e->sil));
          ++e->hist_n;
        }
        ++e->hist_iter;
      }
      if (js_next) {
        int rcj = j_step_local(e, x6d, betas, gt_mm, e->dJraw, js->sqerr, s, nullptr, nullptr, true);
        if (rcj) return rcj;
        rcj = j_step_apply(e, js->J, e->dJraw, js->m, js->v, js->step, js->lr, js->mask, s);
        if (rcj) return rcj;
        reuse_next = js->reuse;
        if (!js->reuse) e->fwd_cached = false;
      }
      continue;
    }
    const bool reuse = !supv && reuse_next && e->fwd_cached && !folded && e->sil_mask == nullptr &&
                       !(e->verts_partial && !(e->have_jsup && e->jsup_fits_known));
    reuse_next = false;
    e->fwd_cached = false;
    prof_mark(e, 0, s);
    // with the pose discriminator its per-joint MLP rides in the chain-forward launch, and its adjoint in the launch of
    // the dF^T slab sum (two independent latency-bound kernels side by side: prep.hip)
    const bool fuse_conv = pd && !reuse;
    // k_tail_step (round 6; prep.hip): the previous iteration's last launch already ran THIS iteration's chain forward (and its per-joint
    // MLP forward, and counted the Adam step) on the poses it had just updated -- `pre_done`
    const bool pre = pre_done && !reuse;
    pre_done = false;
    if (reuse) { if (!e->have_jsup) launch_step_inc(step, s); }      // (with support lists the count rides in k_rejoints_sparse)
    else if (pre) { /* nothing to launch */ }
    else if (fuse_conv) launch_prep_fwd_dconv(e->m, x6d, betas, e->FT, e->FTq, e->AT, e->R0T, e->B, e->BP, step, e->convL, e->H2T, nullptr, s);
    else launch_prep_fwd(e->m, x6d, nullptr, betas, e->FT, e->FTq, e->AT, e->R0T, e->B, e->BP, step, s);
    prof_mark(e, 0, s);
    prof_mark(e, 1, s);
    if (reuse) {
      int rcr = joints_from_stored_verts(e, s, step);
      if (rcr) return rcr;
    } else if (folded) {
      GemmArgs g;   // M^T[(i,j,c)][b] = sum_k H[(i,j,c)][k] F^T[k][b]
      g.A = e->Hk; g.lda = FOLD_M; g.Bm = e->FT; g.ldb = e->BP; g.Out = e->MT; g.ldo = e->BP;
      g.bias = nullptr; g.mask = nullptr; g.split_stride = 0; g.M = FOLD_M; g.N = e->BP; g.K = KFP;
      int rcf = launch_gemm_128x64(g, EPI_STORE, 1, s);
      if (rcf) return rcf;
      launch_fold_fwd(e->MT, e->AT, e->G0, e->Jsum, e->BP, s);
    } else if (supv) {
      ReprojLaunch rls{e->gt_j2d, e->cam, e->gcam, e->sq2d, (float)(2.0 * 0.01 / ((double)e->bnorm * 34.0))};
      int rcs = launch_sup_iter(e->sup, e->sup_nsv, e->Jn_vi, e->FTq, e->AT, gt_mm, jscale, e->joints, sqerr ? sqerr : e->sqerr, e->dA, e->dF,
                                e->B, e->BP, s, e->gt_j2d ? &rls : nullptr);
      if (rcs) return rcs;
    } else {
      const bool silf = e->sil_mask != nullptr;      // the silhouette term needs the vertices
      // (silhouette iterations: the vertices go to the pose-major buffer the rasteriser reads; VTb then only receives the
      // rasteriser's vertex adjoint -- its stored vertices are no longer those of this forward: verts_partial)
      int rcl = launch_lbs_fwd(e->m, e->Jn_vi, e->FTq, e->AT, e->VPb, e->JP, silf ? e->VPM : nullptr, e->B, e->BP, e->nvc, s,
                               e->profiling ? e->probe : nullptr, nullptr, tl, ntl, silf ? 1 : 0);
      if (rcl) return rcl;
    }
    prof_mark(e, 1, s);
    prof_mark(e, 2, s);
    ReprojLaunch rl{e->gt_j2d, e->cam, e->gcam, e->sq2d, (float)(2.0 * 0.01 / ((double)e->bnorm * 34.0))};   // weight 1/100
    if (!supv)
      launch_joints_loss(reuse ? e->dFTp : folded ? e->Jsum : e->JP, reuse ? e->nsplit : folded ? 1 : e->nvc, gt_mm, nullptr, jscale,
                         e->joints, sqerr ? sqerr : e->sqerr, e->dJT, e->B, e->BP, s, e->gt_j2d ? &rl : nullptr, reuse ? 32 : NH,
                         (reuse && e->have_jsup) ? e->jsup.flag : nullptr);
    prof_mark(e, 2, s);
    int rc = 0;
    const bool sil = e->sil_mask != nullptr && !folded;
    if (sil) {   // 100 * mean((silhouette - mask)^2), optimize.py:234-237,252
      prof_mark(e, 8, s);
      const float silscale = (float)(2.0 * 100.0 / ((double)e->bnorm * (double)e->sil * (double)e->sil));
      if (!e->smask_valid) { launch_mask_sq(e->sil_mask, e->smask, e->B, s, e->sil); e->smask_valid = true; }
      // projection, rasterisation, loss and adjoint in one kernel, straight from / into the row-quad vertex buffer
      launch_sil_raster_adj(e->VTb, e->BP, e->cam, e->m.faces_int_pk ? e->m.faces_int_pk : e->m.faces_pk, e->m.nfaces, e->sil_mask, e->smask, e->cover, e->ncover, e->sqsil,
                            silscale, e->gcam, e->gt_j2d ? 1 : 0, e->B, s, e->sil, e->VPM);
      prof_mark(e, 8, s);
    }
    prof_mark(e, 3, s);
    if (folded) launch_fold_bwd(e->dJT, e->AT, e->MT, e->G0, e->dMT, e->dA, e->BP, s);
    else if (!supv) {
      int rcb = launch_lbs_bwd(e->m, e->Jn_iv, e->AT, e->VPb, e->dJT, sil ? e->VTb : nullptr, e->DVP, e->dATp, e->BP, e->nvcb, s, tl, ntl, slab_masks(e));
      if (rcb) return rcb;
    }
    prof_mark(e, 3, s);
    prof_mark(e, 4, s);
    if (folded) {
      GemmArgs g;   // dF^T[k][b] = sum_m H[m][k] dM^T[m][b]   (split over m, partial slabs)
      g.A = e->Hm; g.lda = KFP; g.Bm = e->dMT; g.ldb = e->BP; g.Out = e->dFTp; g.ldo = e->BP;
      g.bias = nullptr; g.mask = nullptr; g.split_stride = (size_t)KFP * e->BP; g.M = KFP; g.N = e->BP; g.K = FOLD_M;
      rc = launch_gemm_224(g, EPI_STORE, e->nsplit, s);
    } else if (!supv) {
      rc = blend_adjoint_gemm(e, s, tl, ntl, ns_adj);
    }
    prof_mark(e, 4, s);
    if (rc) return rc;
    if (pd) {
      prof_mark(e, 5, s);
      rc = disc_forward(e, x6d, nullptr, s, true, fuse_conv);
      if (rc) return rc;
      rc = disc_backward_input(e, x6d, nullptr, nullptr, dscale, 1.f, e->gx, s, e->dsq, !folded);
      prof_mark(e, 5, s);
      if (rc) return rc;
    }
    if (sd) {
      prof_mark(e, 6, s);
      launch_shape_disc(e->Ps, betas, nullptr, e->gb, sscale, 1.f, e->B, s, nullptr, e->ssq);
      prof_mark(e, 6, s);
    }
    prof_mark(e, 7, s);
    // JRR_TAIL_STEP=1 (experiment, round 6: built, parity-green over the whole GPU suite, measured SLOWER -- 1.1355 against 1.1302 ms per
    // iteration): slab sum + k_tail_step instead of the three launches of rounds 2-5 (slab sum || MLP adjoint,
    // k_chain_bwd, the next iteration's chain forward || MLP forward).  Off by default.
    static const bool tail_on = [] { const char* v = getenv("JRR_TAIL_STEP"); return v && v[0] == '1'; }();
    const bool tail = tail_on && !folded && !supv;
    if (folded) launch_reduce_slabs(e->dFTp, e->nsplit, (size_t)KFP * e->BP, e->dF, (size_t)KFP * e->BP, s);
    else if (supv) {      // dA^T / dF^T arrive complete: nothing to sum; the per-joint MLP adjoint runs alone
      if (pd) launch_disc_conv_bwd(e->convL, x6d, e->dH2T, nullptr, dscale, 1.f, e->gx, e->B, e->BP, s, e->dsq, 1);
    } else reduce_adjoint_partials(e, s, (pd && !tail) ? x6d : nullptr, dscale, ns_adj);      // (tail: the MLP adjoint rides in k_tail_step)
    PrepBwdLaunch L;
    L.x6d_in = x6d; L.betas_in = betas;
    if (folded || supv) { L.dATp = e->dA; L.dFTp = e->dF; } else set_adjoint_slabs(e, L);
    L.FT = e->FT; L.R0T = e->R0T; L.AT = e->AT; L.dRT = e->dRT; L.dbT = e->dbT;
    L.gx_extra = pd ? e->gx : nullptr; L.gb_extra = sd ? e->gb : nullptr;
    L.x6d_io = x6d; L.betas_io = betas; L.adam_m = adam_m; L.adam_v = adam_v; L.step = step;
    L.lr = lr; L.B = e->B; L.BP = e->BP;
    if (e->gt_j2d || sil) { L.gcam = e->gcam; L.cam_io = e->cam; L.cam_m = e->cam_m; L.cam_v = e->cam_v; }
    if (tail) {
      // the next iteration is a plain one (no J step in between, which would overwrite F^T / A^T with its own forward): its chain forward
      // [+ per-joint MLP forward] runs at the end of this launch, on the poses Adam has just written
      const bool js_next_ = js && (it + 1) % js->every == 0;
      TailStepLaunch q;
      if (pd) { q.conv_img = e->convL; q.dH2T = e->dH2T; q.dscale = dscale; q.gx = e->gx; q.dsq = e->dsq; }
      q.do_next = (it + 1 < n_iters) && !js_next_;
      q.FTw = e->FT; q.FTq = e->FTq; q.ATw = e->AT; q.R0Tw = e->R0T; q.H2T_next = pd ? e->H2T : nullptr;
      q.step = step; q.arrive = e->step_scratch;
      int rct = launch_tail_step(e->m, q, L, s);
      if (rct) return rct;
      pre_done = q.do_next;
    } else launch_prep_bwd(L, e->m, s);
    prof_mark(e, 7, s);
    if (e->hist) {      // scripts/optimize.py:255-261: the five weighted terms every `hist_every`-th iteration
      if (e->hist_iter % e->hist_every == 0 && e->hist_n < e->hist_cap) {
        hipLaunchKernelGGL(k_loss_record, dim3(1), dim3(1024), 0, s, sqerr ? sqerr : e->sqerr, e->gt_j2d ? e->sq2d : nullptr,
                           sil ? e->sqsil : nullptr, pd ? e->dsq : nullptr, sd ? e->ssq : nullptr, e->B, e->BP, (float)e->bnorm,
                           e->hist + (size_t)e->hist_n * 5, (float)(e->sil * e->sil));
        ++e->hist_n;
      }
      ++e->hist_iter;
    }
    if (js && (it + 1) % js->every == 0) {      // scripts/optimize.py:300-312 inside the call (single process: no collective)
      int rcj = j_step_local(e, x6d, betas, gt_mm, e->dJraw, js->sqerr, s, nullptr, nullptr, true);
      if (rcj) return rcj;
      rcj = j_step_apply(e, js->J, e->dJraw, js->m, js->v, js->step, js->lr, js->mask, s);
      if (rcj) return rcj;
      reuse_next = js->reuse;
      if (!js->reuse) e->fwd_cached = false;
    }
  }
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_refine_run(jrr_engine_t* e, float* x6d, float* betas, const float* gt_mm, float* adam_m,
                              float* adam_v, int32_t* step, float lr, int n_iters, float* sqerr, void* stream) {
  return refine_run_impl(e, x6d, betas, gt_mm, adam_m, adam_v, step, lr, n_iters, sqerr, false, nullptr, stream);
}

extern "C" int jrr_refine_run_after_j_step(jrr_engine_t* e, float* x6d, float* betas, const float* gt_mm, float* adam_m,
                                           float* adam_v, int32_t* step, float lr, int n_iters, float* sqerr, void* stream) {
  return refine_run_impl(e, x6d, betas, gt_mm, adam_m, adam_v, step, lr, n_iters, sqerr, true, nullptr, stream);
}

extern "C" int jrr_refine_run_j_steps(jrr_engine_t* e, float* x6d, float* betas, const float* gt_mm, float* adam_m,
                                      float* adam_v, int32_t* step, float lr, int n_iters, float* sqerr, int j_every,
                                      float* J, float* J_m, float* J_v, int32_t* J_step, float j_lr, const float* mask,
                                      float* j_sqerr, int after_j_step, void* stream) {
  if (!e || j_every <= 0 || !J || !J_m || !J_v || !J_step) { jrr_set_error("refine_run_j_steps: bad argument"); return JRR_ERR_ARG; }
  if (!(e->flags & JRR_FLAG_KEEP_VERTS)) { jrr_set_error("J step requires JRR_FLAG_KEEP_VERTS"); return JRR_ERR_STATE; }
  JStepArgs js{j_every, J, J_m, J_v, J_step, j_lr, mask, j_sqerr, (after_j_step & 2) == 0};
  return refine_run_impl(e, x6d, betas, gt_mm, adam_m, adam_v, step, lr, n_iters, sqerr, (after_j_step & 1) != 0, &js, stream);
}

// =============================================================================================
// J step (scripts/optimize.py:300-312)
// =============================================================================================
// dJn[i][v] = sum over the (plane, pose-split) slabs P[s][i][v]
__global__ void k_djn_reduce(const float* __restrict__ P, int nslab, float* __restrict__ dJn, const int* __restrict__ skip) {
  if (skip && *skip) return;                   // the support-restricted product wrote dJn itself
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NH * VP) return;
  int i = idx / VP, v = idx % VP;
  float acc = 0.f;
  for (int s = 0; s < nslab; ++s) acc += P[((size_t)s * 32 + i) * VP + v];
  dJn[idx] = acc;
}

// dJ from the joint adjoint dJT [3][18][BP] (already in the engine) and the stored vertices VTb [3][VP][BP]
static int j_grad_from_verts(jrr_engine* e, float* dJ, hipStream_t s, float* dJs) {
  if (!(e->flags & JRR_FLAG_KEEP_VERTS)) { jrr_set_error("dJ requires an engine created with JRR_FLAG_KEEP_VERTS"); return JRR_ERR_STATE; }
  if (e->verts_partial && !(e->have_jsup && e->jsup_fits_known)) {
    jrr_set_error("dJ: the stored vertices are those of a J step over the regressor's support; run jrr_find_joints_forward first");
    return JRR_ERR_STATE;
  }
  // over the regressor's support when its lists fit (lbs.hip, "J step over the regressor's SUPPORT"), else the dense product
  const int* sflag = e->have_jsup ? e->jsup.flag : nullptr;
  if (e->have_jsup) launch_jgrad_sparse(e->jsup, e->dJT, e->VTb, e->dJn, e->BP, s);
  if (!(e->have_jsup && e->jsup_fits_known)) {      // (known to fit: the dense product and its slab sum are not even enqueued)
    int rc = launch_jgrad_q(e->dJT, e->VTb, e->dJnp, e->BP, e->nsplitJ, s, sflag);
    if (rc) return rc;
    hipLaunchKernelGGL(k_djn_reduce, dim3((NH * VP + 255) / 256), dim3(256), 0, s, e->dJnp, 3 * e->nsplitJ, e->dJn, sflag);
  }
  launch_jreg_bwd(e->Jraw, e->have_mask ? e->Jmask : nullptr, e->Jn, e->rowsum, e->dJn, VP, dJ, e->m.v2p, s,
                  dJs ? &e->jsup : nullptr, e->m.p2v, dJs);      // dJs: the same gradient on the support lists, same launch
  CHECK_LAUNCH();
  return JRR_OK;
}

extern "C" int jrr_j_regressor_grad(jrr_engine_t* e, const float* x6d, const float* betas, const float* gt_mm,
                                    float* dJ, float* sqerr, float* joints, void* stream) {
  if (!e || !x6d || !betas || !gt_mm || !dJ) return JRR_ERR_ARG;
  if (!e->have_J) { jrr_set_error("J_regressor not set"); return JRR_ERR_STATE; }
  if (!(e->flags & JRR_FLAG_KEEP_VERTS)) { jrr_set_error("J step requires JRR_FLAG_KEEP_VERTS"); return JRR_ERR_STATE; }
  return j_step_local(e, x6d, betas, gt_mm, dJ, sqerr, (hipStream_t)stream, joints);
}

// find_joints on the poses of the J step that preceded, with the CURRENT (stepped) regressor, from that step's stored vertices: the
// joints the driver evaluates after the step (scripts/optimize.py:317-321) without a second SMPL forward
extern "C" int jrr_find_joints_after_j_step(jrr_engine_t* e, const float* x6d, const float* betas, float* joints, void* stream) {
  if (!e || !x6d || !betas || !joints) { jrr_set_error("find_joints_after_j_step: null"); return JRR_ERR_ARG; }
  if (!e->have_J) { jrr_set_error("J_regressor not set"); return JRR_ERR_STATE; }
  const bool ok = e->fwd_cached && e->fc_x6d == x6d && e->fc_betas == betas && e->VTb != nullptr &&
                  !(e->verts_partial && !(e->have_jsup && e->jsup_fits_known));
  if (!ok) {
    jrr_set_error("find_joints_after_j_step: the previous forward on this engine was not a J step (jrr_j_regressor_grad*) on the same pose buffers");
    return JRR_ERR_STATE;
  }
  hipStream_t s = (hipStream_t)stream;
  int rc = joints_from_stored_verts(e, s);
  if (rc) return rc;
  launch_joints_loss(e->dFTp, e->nsplit, nullptr, nullptr, 0.f, joints, nullptr, nullptr, e->B, e->BP, s, nullptr, 32,
                     e->have_jsup ? e->jsup.flag : nullptr);
  CHECK_LAUNCH();
  return JRR_OK;
}

// ---- the J step's all-reduce payload restricted to the regressor's support (include/jrr.h) ----
extern "C" int jrr_j_support_info(jrr_engine_t* e, int32_t* counts_host, int32_t* fits_host, void* stream) {
  if (!e || !fits_host) return JRR_ERR_ARG;
  if (!e->have_jsup || !e->have_J) { jrr_set_error("j_support_info: needs JRR_FLAG_KEEP_VERTS and a regressor"); return JRR_ERR_STATE; }
  int32_t cnt[32] = {0}, flag = 0;
  JRR_HIP(hipStreamSynchronize((hipStream_t)stream));
  JRR_HIP(hipMemcpy(cnt, e->jsup.cnt, NH * sizeof(int32_t), hipMemcpyDeviceToHost));
  JRR_HIP(hipMemcpy(&flag, e->jsup.flag, sizeof(int32_t), hipMemcpyDeviceToHost));
  {   // sticky device error: since the last call the support left the tiles reported then / stopped fitting the lists, while the engine
      // enqueued support-restricted work only (the caller changed J or the mask IN PLACE instead of announcing it through
      // jrr_engine_set_j_regressor): every result since then is suspect.  Cleared by reporting it.
    int32_t err = 0;
    JRR_HIP(hipMemcpy(&err, e->jsup.flag + JSUP_ERR, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (err) {
      JRR_HIP(hipMemset(e->jsup.flag + JSUP_ERR, 0, 2 * sizeof(int32_t)));      // error word and KNOWN
      e->jsup_fits_known = false; e->act_valid = false; e->sup_valid = false; e->fwd_cached = false;
      jrr_set_error("the J_regressor's support GREW behind the engine's back (%s): J or its mask was edited in place after "
                    "jrr_j_support_info; results since then are invalid -- announce a changed regressor with jrr_engine_set_j_regressor",
                    (err & 2) ? "a row no longer fits the support lists" : "entries outside the reported tiles");
      return JRR_ERR_STATE;
    }
  }
  if (counts_host) for (int i = 0; i < NH; ++i) counts_host[i] = cnt[i];
  *fits_host = flag;
  e->jsup_fits_known = flag != 0;      // stays true under J steps (ReLU' = 0: Adam never re-activates an entry); cleared by set_j_regressor
  e->act_valid = false; e->sup_valid = false;
  {   // the baseline the device checks later supports against (k_jsup_tilemask)
    const int32_t known = flag ? 1 : 0;
    if (flag) JRR_HIP(hipMemcpy(e->jsup.tknown, e->jsup.tmask, VT * sizeof(int32_t), hipMemcpyDeviceToDevice));
    JRR_HIP(hipMemcpy(e->jsup.flag + JSUP_KNOWN, &known, sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (flag && (e->flags & JRR_FLAG_SUPPORT_TILES)) {      // the support's tiles, for the kernels of the joint-loss iteration
    int32_t tm[VT], list[VT];
    JRR_HIP(hipMemcpy(tm, e->jsup.tmask, VT * sizeof(int32_t), hipMemcpyDeviceToHost));
    int n = 0;
    for (int t = 0; t < VT; ++t) if (tm[t]) list[n++] = t;
    if (n > 0) {
      JRR_HIP(hipMemcpy(e->act_list, list, n * sizeof(int32_t), hipMemcpyHostToDevice));
      e->nact = n; e->act_valid = true;
    }
    // the support's VERTICES (union of the rows' lists): up to SUP_NSV of them run the per-vertex iteration (supk.h).  J steps only
    // shrink the support, so the set stays a superset; the regressor's values are read live (Jn_vi) at every iteration.
    static const bool sup_off = [] { const char* v = getenv("JRR_SUPPORT_FUSED"); return v && v[0] == '0'; }();
    if (n > 0 && !sup_off) {
      std::vector<int32_t> col((size_t)NH * JSUP_CAP);
      JRR_HIP(hipMemcpy(col.data(), e->jsup.col, col.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
      std::vector<char> seen(VP, 0);
      bool in_range = true;
      for (int i = 0; i < NH; ++i)
        for (int k = 0; k < cnt[i] && k < JSUP_CAP; ++k) {
          const int r = col[(size_t)i * JSUP_CAP + k];
          if (r < 0 || r >= VP) { in_range = false; break; }
          seen[r] = 1;
        }
      std::vector<int32_t> rows;
      for (int r = 0; r < VP; ++r) if (seen[r]) rows.push_back(r);
      if (in_range && !rows.empty() && (int)rows.size() <= SUP_NSV) {
        JRR_HIP(hipMemcpy(e->sup.rows, rows.data(), rows.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        launch_sup_gather(e->m, e->sup, (int)rows.size(), (hipStream_t)stream);
        JRR_HIP(hipStreamSynchronize((hipStream_t)stream));
        e->sup_nsv = (int)rows.size(); e->sup_valid = true;
      }
    }
  }
  return JRR_OK;
}

extern "C" int jrr_engine_support_tiles(const jrr_engine_t* e, int32_t* n_tiles_host) {
  if (!e) return JRR_ERR_ARG;
  const bool on = use_tile_list(e);
  if (n_tiles_host) *n_tiles_host = on ? e->nact : VT;
  return on ? 1 : 0;
}

extern "C" int jrr_engine_support_vertices(const jrr_engine_t* e, int32_t* n_vertices_host) {
  if (!e) return JRR_ERR_ARG;
  const bool on = use_sup_vertices(e);
  if (n_vertices_host) *n_vertices_host = on ? e->sup_nsv : 0;
  return on ? 1 : 0;
}

extern "C" int jrr_j_regressor_grad_support(jrr_engine_t* e, const float* x6d, const float* betas, const float* gt_mm,
                                            float* dJs, float* sqerr, float* joints, void* stream) {
  if (!e || !x6d || !betas || !gt_mm || !dJs) return JRR_ERR_ARG;
  if (!e->have_J) { jrr_set_error("J_regressor not set"); return JRR_ERR_STATE; }
  if (!(e->flags & JRR_FLAG_KEEP_VERTS)) { jrr_set_error("J step requires JRR_FLAG_KEEP_VERTS"); return JRR_ERR_STATE; }
  if (!e->jsup_fits_known) { jrr_set_error("j_regressor_grad_support: call jrr_j_support_info first (it must report fits = 1)"); return JRR_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  return j_step_local(e, x6d, betas, gt_mm, e->dJraw, sqerr, s, joints, dJs, true);
}

extern "C" int jrr_j_step_apply_support(jrr_engine_t* e, float* J, const float* dJs, float* m, float* v, int32_t* step, float lr,
                                        const float* mask, void* stream) {
  if (!e || !J || !dJs || !m || !v || !step) { jrr_set_error("j_step_apply_support: null"); return JRR_ERR_ARG; }
  if (!(e->flags & JRR_FLAG_KEEP_VERTS) || !e->jsup_fits_known) { jrr_set_error("j_step_apply_support: call jrr_j_support_info first (KEEP_VERTS engine, fits = 1)"); return JRR_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  // the dense gradient the optimiser sees: zero outside the support (exactly what the dense path holds there), the
  // all-reduced values on it (scattered inside the update kernel).  Adam itself stays dense: entries that left the support
  // keep coasting on their momentum.
  return j_step_apply(e, J, nullptr, m, v, step, lr, mask, s, dJs);
}

// torch.optim.Adam on the raw regressor with the (all-reduced) gradient, then J*mask -> ReLU -> row-normalise into the
// engine's layouts: the second half of the J step in ONE call (step counter incremented on the device).  The forward
// cached by jrr_j_regressor_grad stays valid: it does not depend on the regressor.
extern "C" int jrr_j_step_apply(jrr_engine_t* e, float* J, const float* dJ, float* m, float* v, int32_t* step, float lr,
                                const float* mask, void* stream) {
  if (!e || !J || !dJ || !m || !v || !step) { jrr_set_error("j_step_apply: null"); return JRR_ERR_ARG; }
  if (!e->has_model) { jrr_set_error("engine was created without an SMPL model (discriminators only)"); return JRR_ERR_STATE; }
  return j_step_apply(e, J, dJ, m, v, step, lr, mask, (hipStream_t)stream);
}

static int j_step_apply(jrr_engine* e, float* J, const float* dJ, float* m, float* v, int32_t* step, float lr, const float* mask,
                        hipStream_t s, const float* dJs) {
  const bool cached = e->fwd_cached, known = e->jsup_fits_known && mask == e->jsup_mask;
  if (!known && e->have_jsup) JRR_HIP(hipMemsetAsync(e->jsup.flag + JSUP_KNOWN, 0, sizeof(int32_t), s));   // (another mask: no baseline to hold the new support against)
  if (e->have_jsup && e->have_J && e->tab_static) {
    // Adam, the engine's copy, the row sums, the normalised layouts and the support lists in ONE launch (lbs.hip k_jstep_update)
    JStepUpdate a;
    a.J = J; a.dJ = dJ; a.dJs = dJs; a.m = m; a.v = v; a.step = step; a.lr = lr;
    a.mask = mask; a.Jraw = e->Jraw; a.Jmask = e->Jmask; a.rowsum = e->rowsum; a.Jn = e->Jn; a.Jn_vi = e->Jn_vi; a.Jn_iv = e->Jn_iv;
    a.Jn_q = e->Jn_q; a.p2v = e->m.p2v; a.v2p = e->m.v2p; a.r16 = (e->m.kjs && e->m.bwd16) ? 1 : 0;
    a.sup = e->jsup; a.sync = e->jsup.flag + 1;
    launch_jstep_update(a, s);
    e->have_mask = mask != nullptr;
    e->jsup_mask = mask;
    e->jsup_fits_known = known;      // the stepped regressor's support is a subset of the old one (ReLU' = 0 outside it; same mask)
    e->fold_valid = false;
    if (e->folded) { int rcf = fold_rebuild(e, s); if (rcf) return rcf; }
    CHECK_LAUNCH();
    return JRR_OK;
  }
  if (!dJ) {      // (no support lists: jrr_j_step_apply_support has refused already; kept for completeness)
    JRR_HIP(hipMemsetAsync(e->dJraw, 0, (size_t)NH * V * sizeof(float), s));
    launch_jsup_scatter(e->jsup, dJs, e->m.p2v, e->dJraw, s);
    dJ = e->dJraw;
  }
  // Adam with step + 1; the counter itself is incremented by the normalisation's first launch (one launch less per J step)
  launch_adam_flat(J, dJ, m, v, (size_t)NH * V, step, lr, 0.9f, 0.999f, 1e-8f, s, 1);
  int rc = set_j_regressor_impl(e, J, mask, (void*)s, step);
  e->fwd_cached = cached;
  e->jsup_fits_known = known;      // the stepped regressor's support is a subset of the old one (ReLU' = 0 outside it; same mask)
  return rc;
}

static int j_step_local(jrr_engine* e, const float* x6d, const float* betas, const float* gt_mm, float* dJ, float* sqerr, hipStream_t s,
                        float* joints, float* dJs, bool support_verts) {
  // v_posed kept: the next inner iteration may reuse this forward.  The vertices (support_verts: the callers whose second half of
  // the step is the engine's own -- the in-call J steps and the support-sized pair): when the regressor's support is known to fit
  // the lists, both consumers (k_jgrad_sparse here, k_rejoints_sparse in the reusing iteration) read support rows only -- the forward
  // stores the tiles that hold one (a few dozen of 216) instead of 340 MB at 4096 poses
  const bool few = support_verts && e->have_jsup && e->jsup_fits_known;
  const bool listed = use_tile_list(e);             // ... and with JRR_FLAG_SUPPORT_TILES nothing but those tiles is computed (any caller: the engine was created for it)
  smpl_forward(e, x6d, nullptr, betas, true, true, nullptr, s, few ? e->jsup.tmask : nullptr, listed ? e->act_list : nullptr, listed ? e->nact : 0);
  e->fwd_cached = true; e->fc_x6d = x6d; e->fc_betas = betas;
  const float scale = (float)(2.0 * 1.0 / ((double)e->bnorm * 51.0));   // optimize.py:307 unweighted MSE
  launch_joints_loss(e->JP, e->nvc, gt_mm, nullptr, scale, joints ? joints : e->joints, sqerr ? sqerr : e->sqerr, e->dJT, e->B, e->BP, s);
  return j_grad_from_verts(e, dJ, s, dJs);
}
