// smi_eng_host.h -- host side of the one-row decode engine (smi_eng.h): the static work plan, the re-packed weight
// stream, create / destroy / launch.  Included by smi_llm.hip after `struct smi_llm`.
#pragma once

namespace {

struct EngGeom {
  int KT[4], NW[4], nparts[4], G[4];
  int nimg[4][4];      // images (MFMAs) of chain set a of a part
  int part_imgs[4];
};

inline EngGeom eng_geom(int H, int Q, int KV, int I, int n_heads) {
  EngGeom g;
  memset(&g, 0, sizeof(g));
  const int NWo = (n_heads == 14 || n_heads == 4) ? n_heads : 8;   // launch_oproj's wave count
  const int KT[4] = {H / 32, Q / 32, H / 32, I / 32}, NW[4] = {16, NWo, 8, 16};
  const int np[4] = {(Q + 2 * KV) / 4, H / 4, 2 * I / 4, H / 4};
  for (int ph = 0; ph < 4; ++ph) {
    g.KT[ph] = KT[ph]; g.NW[ph] = NW[ph]; g.nparts[ph] = np[ph]; g.G[ph] = (NW[ph] + 3) / 4;
    for (int a = 0; a < g.G[ph]; ++a) {
      g.nimg[ph][a] = KT[ph] - 1 - 4 * a >= 0 ? (KT[ph] - 1 - 4 * a) / NW[ph] + 1 : 0;
      g.part_imgs[ph] += g.nimg[ph][a];
    }
  }
  return g;
}

struct EngPlanHost {
  int ncu = 0, maxlen = 0;
  EngGeom g;
  std::vector<EngCuPlan> cu;
  std::vector<uint32_t> desc;   // [ncu][maxlen]: the CU's images of one layer in stream order
  std::vector<uint16_t> lens;   // [ncu]
  int max_wave_phase_imgs = 0, max_slots = 0, max_jobs = 0, min_load = 0, max_load = 0;
};

// Static plan: which CU owns which 4-row parts of the four matrices, which wave runs which (part, chain set) job, and the
// order of the 1-KiB images in every wave's stream.  Deterministic; identical for every layer.  Returns false (with a reason)
// when the model / device does not fit the engine's fixed limits -- the caller then keeps the launch path.
inline bool eng_build_plan(int H, int Q, int KV, int I, int n_heads, int ncu, EngPlanHost& P, char* why, size_t nwhy) {
  auto fail = [&](const char* m) { snprintf(why, nwhy, "%s", m); return false; };
  if (H % 32 || Q % 32 || I % 32 || (Q + 2 * KV) % 16) return fail("dimensions not tile-aligned");
  if (ncu < n_heads + 8 || ncu > 1024) return fail("CU count outside the plan's range");
  P.ncu = ncu;
  P.g = eng_geom(H, Q, KV, I, n_heads);
  const EngGeom& g = P.g;
  for (int ph = 0; ph < 4; ++ph)
    if (g.G[ph] > 4 || g.nparts[ph] > 65535) return fail("too many chains or parts");
  std::vector<std::vector<int>> parts[4];
  for (int ph = 0; ph < 4; ++ph) parts[ph].assign(ncu, {});
  std::vector<long> load(ncu, 0);
  const int head0 = ncu - n_heads;
  // the head CUs run attention instead of o_proj / down_proj and take fewer weight bytes
  for (int c = head0; c < ncu; ++c) load[c] = g.part_imgs[EPH_DOWN] + g.part_imgs[EPH_O];
  // a matrix's parts are spread evenly over the CUs (a phase lasts as long as its busiest CU); among equals the CU with the
  // fewest weight bytes so far takes the part.  HBM time is not what binds -- a layer's share is ~0.1 MB per CU -- so the
  // totals need not be equal.
  auto place = [&](int ph, bool heads_ok) {
    for (int prt = 0; prt < g.nparts[ph]; ++prt) {
      int best = -1;
      for (int c = 0; c < (heads_ok ? ncu : head0); ++c)
        if (best < 0 || parts[ph][c].size() < parts[ph][best].size() ||
            (parts[ph][c].size() == parts[ph][best].size() && load[c] < load[best])) best = c;
      parts[ph][best].push_back(prt);
      load[best] += g.part_imgs[ph];
    }
  };
  place(EPH_DOWN, false);
  place(EPH_O, false);
  place(EPH_GU, true);
  place(EPH_QKV, true);
  P.cu.assign(ncu, EngCuPlan{});
  P.lens.assign((size_t)ncu, 0);
  std::vector<std::vector<uint32_t>> streams((size_t)ncu);
  P.min_load = 1 << 30; P.max_load = 0;
  for (int c = 0; c < ncu; ++c) {
    EngCuPlan& cp = P.cu[c];
    memset(&cp, 0, sizeof(cp));
    cp.head = (int16_t)(c >= head0 ? c - head0 : -1);
    int tot = 0;
    for (int ph = 0; ph < 4; ++ph) {
      cp.pstart[ph] = (uint16_t)tot;
      if ((int)parts[ph][c].size() > kEngMaxSlots) return fail("too many parts of one matrix on a CU");
      if ((int)parts[ph][c].size() > P.max_slots) P.max_slots = (int)parts[ph][c].size();
      for (int prt : parts[ph][c]) {
        if (tot >= kEngMaxParts) return fail("too many parts on a CU");
        cp.parts[tot++] = (uint16_t)prt;
      }
    }
    cp.pstart[4] = (uint16_t)tot;
    if (cp.head >= 0 && (!parts[EPH_O][c].empty() || !parts[EPH_DOWN][c].empty())) return fail("head CU with o_proj / down parts");
    // jobs -> waves: largest jobs first, each to the wave with the fewest images so far
    // (a phase lasts as long as its busiest wave, and a wave's images of one phase must fit its ring: balance per phase,
    // among equals the wave with the fewest images overall)
    std::vector<EngJob> wj[kEngWaves][4];
    int wload[kEngWaves] = {0};
    const int order[4] = {EPH_DOWN, EPH_O, EPH_GU, EPH_QKV};
    for (int oi = 0; oi < 4; ++oi) {
      const int ph = order[oi];
      int pload[kEngWaves] = {0};
      for (int a = 0; a < g.G[ph]; ++a)   // chain sets in order of size (set 0 has the most images): larger jobs first
        for (size_t sl = 0; sl < parts[ph][c].size(); ++sl) {
          int best = 0;   // (the last wave streams the weights and computes the norm factors: no jobs)
          for (int w = 1; w < kEngWaves - 1; ++w)
            if (pload[w] < pload[best] || (pload[w] == pload[best] && wload[w] < wload[best])) best = w;
          EngJob jb;
          jb.part = (uint16_t)parts[ph][c][sl]; jb.set = (uint8_t)a; jb.nimg = (uint8_t)g.nimg[ph][a]; jb.slot = (uint16_t)sl; jb.goff = 0;
          wj[best][ph].push_back(jb);
          pload[best] += g.nimg[ph][a];
          wload[best] += g.nimg[ph][a];
        }
    }
    // the CU's stream of one layer: phase by phase, wave by wave, job by job -- the order the loader issues and the ring holds
    std::vector<uint32_t>& st = streams[(size_t)c];
    int nj[kEngWaves] = {0};
    for (int w = 0; w < kEngWaves; ++w) for (int ph = 0; ph < 5; ++ph) cp.w[w].jstart[ph] = 0;
    // job lists per wave are stored phase-major (jstart), so fill phase by phase
    for (int ph = 0; ph < 4; ++ph) {
      int pimg_cu = 0;
      for (int w = 0; w < kEngWaves; ++w) {
        EngWavePlan& wp = cp.w[w];
        wp.jstart[ph] = (uint16_t)nj[w];
        int pimg = 0;
        for (EngJob jb : wj[w][ph]) {
          if (nj[w] >= kEngMaxJobs) return fail("too many jobs on a wave");
          if (st.size() + jb.nimg > 65535) return fail("stream too long");
          jb.goff = (uint16_t)st.size();
          wp.jobs[nj[w]++] = jb;
          for (int r = 0; r < jb.nimg; ++r) st.push_back(((uint32_t)ph << 30) | ((uint32_t)jb.set << 26) | ((uint32_t)r << 16) | jb.part);
          pimg += jb.nimg;
        }
        wp.pimg[ph] = (uint16_t)pimg;
        if (pimg > P.max_wave_phase_imgs) P.max_wave_phase_imgs = pimg;
        pimg_cu += pimg;
      }
      cp.pcount[ph] = (uint16_t)pimg_cu;
      if (pimg_cu > kEngRingTotal) return fail("a CU's images of one phase exceed its ring");
    }
    for (int w = 0; w < kEngWaves; ++w) {
      cp.w[w].jstart[4] = (uint16_t)nj[w];
      if (nj[w] > P.max_jobs) P.max_jobs = nj[w];
    }
    cp.len_cu = (uint16_t)st.size();
    P.lens[(size_t)c] = (uint16_t)st.size();
    if ((int)st.size() > P.maxlen) P.maxlen = (int)st.size();
    const long cl = (long)st.size();
    if (cl < P.min_load) P.min_load = (int)cl;
    if (cl > P.max_load) P.max_load = (int)cl;
  }
  if (P.maxlen == 0) return fail("empty plan");
  P.desc.assign((size_t)ncu * P.maxlen, 0);
  for (size_t cw = 0; cw < streams.size(); ++cw)
    for (size_t i = 0; i < streams[cw].size(); ++i) P.desc[cw * P.maxlen + i] = streams[cw][i];
  return true;
}

struct EngState {
  int enabled = 0, ncu = 0, maxlen = 0, lds = 0, gran_per_buf = 0;
  EngCuPlan* plan = nullptr;
  unsigned char* stream = nullptr;
  smi_u64* gran = nullptr;
  unsigned* words = nullptr;          // [0] serial, [4..7] err, [8] arrive
  unsigned long long* stamps = nullptr;
  unsigned timeout_ticks = 0;
  int ld_burst = 16, ld_sleep = 0, poll_quiet = 1;
  int edge_delay[5] = {16, 16, 16, 16, 16};
  char why[160] = "";
};

}  // namespace
