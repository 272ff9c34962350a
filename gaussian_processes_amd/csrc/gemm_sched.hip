// XCD-aware data-parallel schedule for the large 128-tile GEMM launches.
//
// MI355X has 8 XCDs with a private 4 MiB L2 each; workgroups are dealt to the XCDs round-robin
// by block id and each XCD keeps 64 of this kernel's workgroups resident (32 CUs x 2), refilled in
// id order.  So the ids = x (mod 8) form XCD x's private in-order queue.  A tile walk that balances
// triangular k ranges by going down the diagonals (heavy tiles first) puts 64 tiles with 64
// different row panels and 64 different column panels on an XCD: nothing is shared in L2 and every
// workgroup streams its operands over the fabric (11.9 GB per N^3/3 launch at N = 8192, against
// 0.8 GB compulsory).  Here the output is cut into macro-tiles of G x G tiles (G = 8: 8 + 8 panels
// serve 64 tiles), the macro-tiles are distributed over the eight queues by longest-processing-time
// first on their k-step totals, light macro-tiles being split into quarters for the final
// balancing, and each queue is written to the ids = x (mod 8) of a tile table the kernel reads
// (gemm.hip).  Placement is an assumption about the dispatcher that affects speed only: every tile
// is computed exactly once whatever the hardware does with the ids.
#include "gemm_core.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace gpfit {

namespace {
struct Plan {
  int* table = nullptr;  // device
  int blocks = 0;
};
using Key = std::tuple<int, int, int, int, int, int, int, int>;  // device, M, N, K, lower, a_tri, b_tri, K step
std::map<Key, Plan> g_plans;
std::mutex g_mutex;

struct Item {
  long weight;
  std::vector<int> tiles;  // packed (ti << 16 | tj), in issue order
};
}  // namespace

template <typename R>
static int build(const GemmArgsT<R>& a, Plan& plan) {
  constexpr int KT = 128 / (int)sizeof(R);
  const int tm = a.M / TILE, tn = a.N / TILE;
  auto ksteps = [&](int ti, int tj) {
    int kb = 0, ke = a.K;
    if (a.a_tri == 1) ke = std::min(ke, ti * TILE + TILE);
    if (a.a_tri == 2) kb = std::max(kb, ti * TILE);
    if (a.b_tri == 1) kb = std::max(kb, tj * TILE);
    if (a.b_tri == 2) ke = std::min(ke, tj * TILE + TILE);
    return std::max(0, ke - kb) / KT;
  };
  // tuning knobs: macro-tile rows x columns, and the weight (as a fraction 1/cutdiv of a queue's
  // average load) below which a macro-tile is split into quarters
  static const int gr_env = getenv("GPFIT_XCD_GR") ? atoi(getenv("GPFIT_XCD_GR")) : 2;
  static const int gc_env = getenv("GPFIT_XCD_GC") ? atoi(getenv("GPFIT_XCD_GC")) : 2;
  static const int cutdiv = getenv("GPFIT_XCD_CUT") ? atoi(getenv("GPFIT_XCD_CUT")) : 8;
  const int GR = std::max(1, std::min(gr_env, 32)), GC = std::max(1, std::min(gc_env, 32));
  // macro-tiles; every tile carries one k step of fixed cost (prologue / epilogue) in its weight
  auto make_items = [&](int I0, int I1, int J0, int J1, std::vector<Item>& out) {
    Item it;
    it.weight = 0;
    for (int ti = I0; ti < std::min(I1, tm); ++ti)
      for (int tj = J0; tj < std::min(J1, tn); ++tj) {
        if (a.out_lower && tj > ti) continue;
        it.tiles.push_back((ti << 16) | tj);
        it.weight += ksteps(ti, tj) + 2;
      }
    if (!it.tiles.empty()) out.push_back(std::move(it));
  };
  std::vector<Item> items;
  for (int I = 0; I < tm; I += GR)
    for (int J = 0; J < tn; J += GC) {
      if (a.out_lower && J >= I + GR) continue;
      make_items(I, I + GR, J, J + GC, items);
    }
  long total = 0;
  for (auto& it : items) total += it.weight;
  // split what is light against the per-queue average into quarters (finer final balancing)
  const long cut = cutdiv > 0 ? total / 8 / cutdiv : 0;
  std::vector<Item> fine;
  for (auto& it : items) {
    if (it.weight > cut || GR < 2 || GC < 2) { fine.push_back(std::move(it)); continue; }
    const int ti0 = (it.tiles.front() >> 16) / GR * GR, tj0 = (it.tiles.front() & 0xffff) / GC * GC;
    const int hr = (GR + 1) / 2, hc = (GC + 1) / 2;
    make_items(ti0, ti0 + hr, tj0, tj0 + hc, fine);
    make_items(ti0, ti0 + hr, tj0 + hc, tj0 + GC, fine);
    make_items(ti0 + hr, ti0 + GR, tj0, tj0 + hc, fine);
    make_items(ti0 + hr, ti0 + GR, tj0 + hc, tj0 + GC, fine);
  }
  std::stable_sort(fine.begin(), fine.end(), [](const Item& x, const Item& y) { return x.weight > y.weight; });
  std::vector<int> queue[8];
  long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (auto& it : fine) {
    int best = 0;
    for (int x = 1; x < 8; ++x)
      if (load[x] < load[best]) best = x;
    load[best] += it.weight;
    queue[best].insert(queue[best].end(), it.tiles.begin(), it.tiles.end());
  }
  size_t len = 0;
  for (auto& q : queue) len = std::max(len, q.size());
  std::vector<int> table(8 * len, -1);
  for (int x = 0; x < 8; ++x)
    for (size_t i = 0; i < queue[x].size(); ++i) table[8 * i + x] = queue[x][i];
  plan.blocks = (int)table.size();
  GP_HIP(hipMalloc((void**)&plan.table, table.size() * sizeof(int)));
  GP_HIP(hipMemcpy(plan.table, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}

template <typename R>
bool gemm_xcd_applies(const GemmArgsT<R>& a) {
  if ((a.M % TILE) || (a.N % TILE) || a.split_k > 1 || a.batch > 1 || a.nptr > 0 || (a.tile && a.tile != TILE)) return false;
  if (a.out_lower && a.M != a.N) return false;
  const long tm = a.M / TILE, tn = a.N / TILE;
  if (tm >= 32768 || tn >= 32768) return false;
  // A data-parallel schedule needs several rounds of the 512 resident workgroups to balance; below
  // that the stream-K / heavy-first walks win (N = 4096: 528 tiles, 7.27 vs 7.5 ms per fit).
  static const long min_tiles = getenv("GPFIT_XCD_MIN_TILES") ? atol(getenv("GPFIT_XCD_MIN_TILES")) : 1536;
  return (a.out_lower ? tm * (tm + 1) / 2 : tm * tn) >= min_tiles;
}
template bool gemm_xcd_applies<double>(const GemmArgsT<double>&);
template bool gemm_xcd_applies<float>(const GemmArgsT<float>&);

template <typename R>
int launch_gemm_xcd(const GemmArgsT<R>& a, hipStream_t s) {
  if (!gemm_xcd_applies(a)) return 1;
  int device = 0;
  GP_HIP(hipGetDevice(&device));
  Plan plan;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    const Key key{device, a.M, a.N, a.K, a.out_lower, a.a_tri, a.b_tri, (int)sizeof(R)};
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
      Plan np;
      const int rc = build(a, np);
      if (rc != 0) return rc;
      it = g_plans.emplace(key, np).first;
    }
    plan = it->second;
  }
  GemmArgsT<R> p = a;
  p.tile = TILE;
  p.sched = plan.table;
  p.sched_blocks = plan.blocks;
  p.reverse &= 4;  // the table fixes the tile walk; only the k direction bit survives
  return launch_gemm_plain(p, s);
}

template int launch_gemm_xcd<double>(const GemmArgsT<double>&, hipStream_t);
template int launch_gemm_xcd<float>(const GemmArgsT<float>&, hipStream_t);

}  // namespace gpfit
