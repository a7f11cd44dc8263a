// Static schedule of the fused dense-block kernel (rdb_fused_bf16_kernel, conv_bf16.hip): plain constexpr C++17, no HIP — the kernel
// includes it, and tests/test_fused_schedule.py compiles it with g++ and replays the schedule against an independent model of the
// issue order (what each counted wait guarantees, ring-slot and tile-buffer lifetimes).
// (no include guard: conv_bf16.hip includes it once per tile height through fused_block.inc)
#ifndef SR_FZ_NS
#define SR_FZ_NS fz
#endif
#ifndef SR_FZ_PT
#define SR_FZ_PT 2
#endif
#ifndef SR_FZ_NW
#define SR_FZ_NW 8  // waves of a workgroup (each owns SR_FZ_PT tile rows): 8 = two per SIMD, 256 registers each; 4 = one per SIMD, 512
#endif
namespace SR_FZ_NS {
constexpr int NW = SR_FZ_NW, GPW = 8 / NW;  // GPW: LDS-DMA instructions of one wave per weight group (a group = eight 1 KB pieces)
static_assert(NW * GPW == 8, "a weight group is dealt out evenly");
#ifndef SR_FZ_SLICED
#define SR_FZ_SLICED (SR_FZ_NW == 4)
#endif
// one wave per SIMD: nobody else's MFMAs cover a wave's address arithmetic, so what a step issues to memory is dealt out over the gaps
// behind ALL of its MFMAs (same order: the counted waits do not change)
constexpr bool kSliced = SR_FZ_SLICED;
constexpr int PT = SR_FZ_PT, TH = NW * PT, XROW = 34, XPIX = (TH + 2) * XROW;
constexpr int XU = (XPIX * 32 + 1023) / 1024, XBUF = XU * 1024, NTB = 6;  // 1 KiB pieces / bytes of a tile buffer ((TH + 2) x 34 pixels x 32 B, rounded up: 20 / 11)
constexpr int TPW = (2 * XU + NW - 1) / NW;  // pieces of a tile pair (two chunks) per wave: 5 (exactly 40 pieces) / 3 (22 pieces + 2 into a spare KB)
constexpr bool kTileSurplus = TPW * NW > 2 * XU;
#ifndef SR_FZ_RING
#define SR_FZ_RING 36
#endif
constexpr int RING = SR_FZ_RING;                    // weight ring, pieces (36 beside six 20 KB tile buffers; the 8-row instance has room for more)
constexpr int LDS_W0 = 0;                           // the ring first: its reads then need no address arithmetic (16-bit offsets)
constexpr int LDS_X0 = RING * 1024;                 // tile buffers
constexpr int LDS_FLAGS = LDS_X0 + NTB * XBUF;      // 64 words: neighbour progress words land here
constexpr int LDS_BIAS = LDS_FLAGS + 256;           // 5 x 64 floats
constexpr int LDS_WTAB = LDS_BIAS + 5 * 64 * 4;     // source offset of every weight piece (480 words)
constexpr int LDS_SPARE = LDS_WTAB + 480 * 4;       // 1 KB for the surplus pieces of a tile pair (8-row tiles only)
// Lean instances (SR_FZ_HALO): a conv's tile goes from the accumulators straight into the LDS buffer of the input it becomes (its
// producer is the workgroup that reads it); only the one-pixel ring around it comes from the neighbours through memory — the tile's
// first / last two 1 KB pieces in place (rows 0 and TH + 1 and what shares their pieces: re-fetched pixels of the workgroup's own,
// already published, border rows) and the two halo COLUMNS through 1 KB of staging per chunk (a DMA's LDS destination is contiguous;
// a wave copies them to their places a step before the first use).  HPW in-place pieces per wave; waves 0 / 1 add the staging piece
// of chunk 0 / 1 right behind theirs (the counted waits are those of a wave without it: a wave with one more operation in flight
// only waits longer).
#ifndef SR_FZ_HALO
#define SR_FZ_HALO 1
#endif
constexpr bool kHalo = SR_FZ_HALO;
constexpr int HPW = 8 / NW;
constexpr int LDS_END0 = LDS_SPARE + (kTileSurplus ? 1024 : 0);
constexpr int LDS_STAGE = LDS_END0 + 2048 <= 160 * 1024 ? LDS_END0 : LDS_WTAB;  // (the lean instances do not use the offset table)
constexpr int LDS_BYTES = LDS_END0 > LDS_STAGE + 2048 ? LDS_END0 : LDS_STAGE + 2048;
static_assert(LDS_BYTES <= 160 * 1024, "fused dense block: LDS");
static_assert(LDS_STAGE >= LDS_END0 || (LDS_STAGE == LDS_WTAB && !kTileSurplus), "staging must not share bytes with the spare KB");
constexpr int NG = 6;  // accumulator groups: conv1..conv4, conv5 couts 0-31 / 32-63
constexpr int MAXSTEPS = 80, MAXP = 480;
constexpr int SC1 = 16;  // cache-policy bit of the buffer builtins: agent scope
#ifndef SR_FZ_AR
#define SR_FZ_AR 4
#endif
constexpr int AR = SR_FZ_AR;  // ring of weight fragments in registers: read AR-1 fragments ahead of their MFMAs
static_assert(AR == 3 || AR == 4, "fragment ring: 3 and 4 are measured (equal); a build with 5 faults at launch (round 4), 6 fails the schedule");

struct StepD {
  int in, chunk, g0, ng, dx0, ndx, tb;
  int wp0, wpn;     // weight pieces [wp0, wp0 + wpn) of the stream, order (tap column, group, tap row)
  int post;         // k in 1..5: conv k is complete after this step (epilogue)
  int K;            // vector-memory instructions of this wave that may still be in flight when the step starts
  int q0, q1;       // weight piece groups [q0, q1) issued after the step's barrier
  int publish;      // k in 1..4: conv k's tile is published after the step's barrier (its stores are covered by K)
  int tile_in;      // > 0: the tile of input group tile_in is issued after the step's barrier (neighbour flags inspected before it)
  int flag_in;      // > 0: wave 0 fetches the neighbour flags for input group flag_in during this step
  int Kflag;        // tile_in > 0: instructions younger than that flag fetch
  int zero_mask;    // accumulator groups that start with this step
  int first_of_in;  // > 0: first step that reads input group first_of_in
  int uc0, dc0;     // units / tap columns before this step (parities select the operand registers)
  int pre;          // the operands of this step's first unit were read during the previous step
  int nx_tile;      // 1 / 2: the next round's x chunks 0-1 / 2-3 are issued after the step's barrier (their buffers are free)
  int nx_q0, nx_q1; // the next round's weight piece groups [nx_q0, nx_q1) are issued after the step's barrier
  int Kclaim;       // claim == 2: thread 0's instructions younger than its ticket atomic
  int copy_in;      // > 0 (lean instances): the halo columns of input group copy_in go from staging to their places behind this step's barrier
  int claim;        // 1: the workgroup's next tile is claimed during this step (its last neighbour flags have been seen), 2: the
                    // ticket is handed to all waves (through the LDS; readable behind the next barrier)
};
struct Sched {
  int nsteps, npieces, ngroups, q_init, q_ahead, ok;  // q_ahead: groups of the next round issued during a round (the rest, up to q_init, at its start)
  StepD st[MAXSTEPS];
  unsigned wtab[MAXP];  // per piece: conv (4 bits) << 28 | byte offset inside that conv's packed image
};

// phases in execution order: {input group, first chunk, end chunk, first accumulator group, groups, one step per tap column (else per
// chunk), conv completed}.  After conv k's own phase come partial sums that do not depend on x_k, ~144 MFMAs per wave and hand-off.
constexpr int kNPhase = 13;
#ifndef SR_FZ_PERDX
#define SR_FZ_PERDX 1  // 1: the partial-sum phases advance one tap column per step (62 steps per tile); 0: one chunk per step (34; conv5's last input stays per column: the next tile's read-ahead needs those steps)
#define SR_FZ_CLAIMLEAD 3  // steps between the ticket atomic and its hand-over
#endif
// (SR_FZ_PERDX2: the same switch for the partial-sum phases of TWO accumulator groups only — 18 weight pieces per chunk, two steps'
// worth fit a ring of 36 — while the three-group phases, 27 pieces per chunk, keep a tap column per step; default = SR_FZ_PERDX)
#ifndef SR_FZ_PERDX2
#define SR_FZ_PERDX2 SR_FZ_PERDX
#endif
#ifndef SR_FZ_PERDX2B
#define SR_FZ_PERDX2B SR_FZ_PERDX2
#endif
constexpr int kPhase[kNPhase][7] = {{0, 0, 4, 0, 1, 0, 1},                                                          // conv1 x
                                    {0, 0, 4, 1, 2, SR_FZ_PERDX2, 0}, {0, 0, 2, 3, 3, SR_FZ_PERDX, 0},              //   conv2-3 x | conv4-5 x[0,1]
                                    {1, 0, 2, 1, 1, 0, 2},                                                          // conv2 x1
                                    {0, 2, 4, 3, 3, SR_FZ_PERDX, 0}, {1, 0, 2, 2, 1, 0, 0},                         //   conv4-5 x[2,3] | conv3 x1
                                    {2, 0, 2, 2, 1, 0, 3},                                                          // conv3 x2
                                    {1, 0, 2, 3, 3, SR_FZ_PERDX, 0}, {2, 0, 2, 3, 1, 0, 0},                         //   conv4-5 x1 | conv4 x2
                                    {3, 0, 2, 3, 1, 0, 4},                                                          // conv4 x3
                                    {2, 0, 2, 4, 2, SR_FZ_PERDX2B, 0}, {3, 0, 2, 4, 2, SR_FZ_PERDX2B, 0},           //   conv5 x2 | conv5 x3
                                    {4, 0, 2, 4, 2, 1, 5}};                                               // conv5 x4
#ifndef SR_FZ_PUBLAG
#define SR_FZ_PUBLAG {4, 2, 2, 3}
#define SR_FZ_TILELAG {5, 4, 4, 5}
#define SR_FZ_FLAGLEAD {4, 3, 3, 4}
#endif
constexpr int kPubLag[4] = SR_FZ_PUBLAG;      // steps from conv k's epilogue to its publish
constexpr int kTileLag[4] = SR_FZ_TILELAG;    // steps from the publish to the fetch of the dependent tile
constexpr int kFlagLead[4] = SR_FZ_FLAGLEAD;  // the flags are fetched this many steps before they are inspected
constexpr int in_chunks(int s) { return s == 0 ? 4 : 2; }
constexpr int in_cb0(int s) { return s == 0 ? 0 : 4 + 2 * (s - 1); }
constexpr int in_tb0(int s) { return s == 0 ? 0 : s == 1 ? 4 : s == 2 ? 0 : s == 3 ? 2 : 4; }

// conv2 completes while 160 of 256 registers are accumulators: its mask is fetched last-minute (the four-wave instance has room: a step ahead)
constexpr int kMaskLead[4] = {1, NW == 4 ? 1 : 0, 1, 1};
constexpr int mask_conv_at(const Sched& s, const int i) {  // the conv (1..4) whose mask is fetched at step i, or 0
  for (int k = 1; k <= 4; ++k) {
    const int j = i + kMaskLead[k - 1];
    if (j < s.nsteps && s.st[j].post == k) return k;
  }
  return 0;
}
constexpr Sched make_sched(const int mode) {  // 0 generic epilogues, 1 lean forward, 2 lean transposed block (they differ in K / Kflag only)
  Sched s{};
  int ns = 0, np = 0, seen = 0, uc = 0, dc = 0;
  int post_step[6] = {-1, -1, -1, -1, -1, -1}, first_use[5] = {-1, -1, -1, -1, -1};
  for (int ph = 0; ph < kNPhase; ++ph) {
    const int in = kPhase[ph][0], g0 = kPhase[ph][3], ng = kPhase[ph][4], perdx = kPhase[ph][5];
    for (int c = kPhase[ph][1]; c < kPhase[ph][2]; ++c)
      for (int part = 0; part < (perdx ? 3 : 1); ++part) {
        StepD& d = s.st[ns];
        d.in = in;
        d.chunk = c;
        d.g0 = g0;
        d.ng = ng;
        d.dx0 = perdx ? part : 0;
        d.ndx = perdx ? 1 : 3;
        d.tb = in_tb0(in) + c;
        d.wp0 = np;
        d.wpn = d.ndx * ng * 3;
        d.uc0 = uc;
        d.dc0 = dc;
        uc += d.ndx * ng;
        dc += d.ndx;
        for (int j = 0; j < d.wpn; ++j) {
          const int dx = d.dx0 + j / (ng * 3), g = g0 + (j / 3) % ng, dy = j % 3;
          const int k = g < 4 ? g : 4, cot = g < 4 ? 1 : 2, cc = g < 4 ? 0 : g - 4;
          const int cb = in_cb0(in) + c;
          s.wtab[np + j] = ((unsigned)k << 28) | (unsigned)((cb * 9 * cot + (dy * 3 + dx) * cot + cc) * 1024);
        }
        np += d.wpn;
        for (int g = g0; g < g0 + ng; ++g)
          if (!(seen >> g & 1)) {
            d.zero_mask |= 1 << g;
            seen |= 1 << g;
          }
        if (first_use[in] < 0) {
          first_use[in] = ns;
          d.first_of_in = in;
        }
        ++ns;
      }
    if (kPhase[ph][6]) {
      s.st[ns - 1].post = kPhase[ph][6];
      post_step[kPhase[ph][6]] = ns - 1;
    }
  }
  s.nsteps = ns;
  s.npieces = np;
  // not across an epilogue (register pressure), not into the first step of x1..x4 (its halo columns reach their places during the step before)
  for (int i = 1; i < ns; ++i) s.st[i].pre = (s.st[i - 1].post || s.st[i].first_of_in > 0) ? 0 : 1;
  while (np % 8) {  // whole groups of eight: the padding re-loads piece 0 into a free slot
    s.wtab[np] = s.wtab[0];
    ++np;
  }
  s.ngroups = np / 8;
  s.ok = 1;
  int issue_step[5] = {-1, 0, 0, 0, 0}, last_use[5] = {0, 0, 0, 0, 0};
  for (int j = 0; j < ns; ++j) last_use[s.st[j].in] = j;
  for (int t = 1; t <= 4; ++t) {  // conv t -> input group t
    const int pub = post_step[t] + kPubLag[t - 1], ti = pub + kTileLag[t - 1], tf = ti - kFlagLead[t - 1];
    if (post_step[t] < 0 || tf <= pub || ti >= first_use[t] - 1) s.ok = 0;
    s.st[pub].publish = t;
    s.st[ti].tile_in = t;
    s.st[tf].flag_in = t;
    issue_step[t] = ti;
    if (first_use[t] >= 1) s.st[first_use[t] - 1].copy_in = t;  // (ti < first_use - 1: the barrier of that step says the tile has landed)
    if (t > 1 && ti <= first_use[t - 1] - 1) s.ok = 0;  // the staging KB still holds the previous input's columns
  }
  // a tile buffer is never refilled while its previous content is still read: last read of the old input < issue step of the new one
  for (int t = 1; t <= 4; ++t)
    for (int j = 0; j < ns; ++j) {
      const bool same_buf = s.st[j].tb == in_tb0(t) || s.st[j].tb == in_tb0(t) + 1;
      if (same_buf && s.st[j].in != t && j >= issue_step[t] && j <= last_use[t]) s.ok = 0;  // somebody else reads it while t owns it
      if (same_buf && s.st[j].in == t && j < issue_step[t]) s.ok = 0;
      // lean instances: conv t's epilogue writes its tile into these buffers (step post_step[t]): nobody else may still read them
      if (same_buf && s.st[j].in != t && j >= post_step[t] && j <= last_use[t]) s.ok = 0;
    }
  // read-ahead for the next round (always issued; behind the last round it fetches nothing useful): x chunks as soon as their
  // buffers hold nothing that is still read, the first weight groups as soon as their ring slots do
  {
    int last01 = 0, last23 = 0;
    for (int j = 0; j < ns; ++j) {
      if (s.st[j].tb <= 1) last01 = j;
      if (s.st[j].tb == 2 || s.st[j].tb == 3) last23 = j;
    }
    if (last01 + 1 >= ns || last23 + 1 >= ns || last01 >= last23) s.ok = 0;
    // the next tile is claimed once this one depends on nobody any more (the step that issues x4's tile has seen the last neighbour
    // flags), the ticket reaches all waves a few steps later; its x chunks follow as soon as ticket and buffers are there
    int claim_at = -1, hand_at = -1;
    for (int j = 0; j < ns; ++j) {
      if (s.st[j].tile_in == 4) claim_at = j;
      if (s.st[j].first_of_in == 4) hand_at = j;
    }
    if (claim_at < 0 || hand_at - SR_FZ_CLAIMLEAD < claim_at || hand_at + 2 >= ns) s.ok = 0;
    claim_at = hand_at - SR_FZ_CLAIMLEAD;  // (not earlier than necessary: the ticket waits in a register of thread 0 until the hand-over)
    s.st[claim_at].claim = 1;
    s.st[hand_at].claim = 2;
    const int n1 = last01 + 1 > hand_at + 1 ? last01 + 1 : hand_at + 1;
    const int n2 = last23 + 1 > n1 + 1 ? last23 + 1 : n1 + 1;
    if (n2 >= ns) s.ok = 0;
    s.st[n1].nx_tile = 1;
    s.st[n2].nx_tile = 2;
    const int total = s.ngroups * 8;
    int k = 0;
    for (int i = 0; i < ns && k < RING / 8; ++i) {
      s.st[i].nx_q0 = s.st[i].nx_q1 = k;
      for (;;) {  // group k = slots [8k, 8k+8): free once their last occupants of this round are consumed (read by a step < i)
        if (k >= RING / 8) break;
        bool free_now = true;
        for (int j = 8 * k; j < 8 * k + 8; ++j) {
          const int occ = j + RING * ((total - 1 - j) / RING);
          if (occ < s.npieces && occ >= s.st[i].wp0) free_now = false;
          // the slot's occupant must also have been ISSUED before this barrier (its DMA is then older than the new one)
          if (occ >= s.st[i].wp0 + RING - 7) free_now = false;
        }
        if (!free_now) break;
        ++k;
      }
      s.st[i].nx_q1 = k;
    }
    s.q_ahead = k;
  }
  // the issue sequence of one wave: what is younger than the operations a step needs may stay in flight at its wait
  int cseq = 0;
  int seq = 0, gend[MAXP / 8] = {}, tend[5][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}}, send[5] = {0, 0, 0, 0, 0}, fseq[5] = {0, 0, 0, 0, 0};
  seq += TPW;  // chunks 0-1 of x: 2 XU pieces = TPW per wave
  tend[0][0] = seq;
  seq += TPW;  // chunks 2-3
  tend[0][1] = seq;
  int issued = RING / 8;  // whole groups that fit the empty ring
  s.q_init = issued;
  for (int q = 0; q < issued; ++q) gend[q] = seq += GPW;
  for (int i = 0; i < ns; ++i) {
    StepD& d = s.st[i];
    // the barrier of step i says: step i's operands have landed, and so have those that step i reads ahead for step i+1 — its tile and
    // its first AR-1 weight fragments
    int need = 0;
    for (int j = i; j <= i + 1 && j < ns; ++j) {
      const StepD& dj = s.st[j];
      const int qn = (j == i ? dj.wp0 + dj.wpn - 1 : dj.wp0 + AR - 2) / 8;
      const int te = tend[dj.in][dj.chunk / 2];
      if (gend[qn] == 0 || te == 0 || dj.wpn > RING || dj.wpn < AR - 1) s.ok = 0;
      if (gend[qn] > need) need = gend[qn];
      if (te > need) need = te;
    }
    if (d.publish) {
      if (send[d.publish] == 0) s.ok = 0;
      if (send[d.publish] > need) need = send[d.publish];
    }
    d.K = seq - need;
    if (d.claim == 2) d.Kclaim = seq - cseq;  // (handed over before this step issues anything)
    if (d.tile_in) {
      d.Kflag = seq - fseq[d.tile_in];
      seq += (mode && kHalo) ? HPW : TPW;  // two chunks: whole (generic instance) or their ring
      tend[d.tile_in][0] = seq;
    }
    int q1 = (d.wp0 + RING) / 8;
    if (q1 > s.ngroups) q1 = s.ngroups;
    d.q0 = issued;
    d.q1 = q1;
    for (int q = issued; q < q1; ++q) gend[q] = seq += GPW;
    issued = q1;
    if (d.nx_tile) seq += TPW;
    seq += (d.nx_q1 - d.nx_q0) * GPW;
    if (mode && d.first_of_in == 4) seq += 8 * PT;  // lean kernels: conv5's residual sources are fetched here (2 sources x 2 cout tiles x PT rows x 2 blocks)
    if (mode == 2 && mask_conv_at(s, i)) seq += 2 * PT;  // ... and conv k's mask kMaskLead steps before its epilogue
    if (d.claim == 1) cseq = seq;  // thread 0 only: the ticket atomic, behind everything this step issues
    if (d.flag_in) fseq[d.flag_in] = seq;  // wave 0 only: one more instruction right here (not counted: the other waves' waits get stricter)
    if (d.post >= 1 && d.post <= 4) {
      seq += 2 * PT;  // the epilogue's stores: PT rows x 2 channel blocks
      send[d.post] = seq;
    }
    // (the counter has six bits; a smaller number only waits for more)
    if (d.K > 60) d.K = 60;
    if (d.Kflag > 60) d.Kflag = 60;
    if (d.Kclaim > 60) d.Kclaim = 60;
  }
  if (issued != s.ngroups) s.ok = 0;
  return s;
}
constexpr Sched kS = make_sched(0), kSL = make_sched(1), kSB = make_sched(2);
#ifndef SR_FZ_PROBE
static_assert(kS.ok == 1 && kSL.ok == 1 && kSB.ok == 1 && kS.nsteps <= MAXSTEPS && kS.npieces == 468, "fused dense block schedule");
#endif
template <int MODE, int S>
constexpr int step_K() { return MODE == 0 ? kS.st[S].K : MODE == 1 ? kSL.st[S].K : kSB.st[S].K; }
template <int MODE, int S>
constexpr int step_Kflag() { return MODE == 0 ? kS.st[S].Kflag : MODE == 1 ? kSL.st[S].Kflag : kSB.st[S].Kflag; }
template <int MODE, int S>
constexpr int step_Kclaim() { return MODE == 0 ? kS.st[S].Kclaim : MODE == 1 ? kSL.st[S].Kclaim : kSB.st[S].Kclaim; }
}  // namespace SR_FZ_NS
