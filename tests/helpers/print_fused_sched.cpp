// Prints the static schedule of the fused dense-block kernel (csrc/fused_sched.h) as JSON, one object per epilogue mode.
#include <cstdio>
#include "fused_sched.h"
using namespace SR_FZ_NS;
static void dump(const Sched& s, int mode) {
  printf("{\"mode\": %d, \"ok\": %d, \"nsteps\": %d, \"npieces\": %d, \"ngroups\": %d, \"q_init\": %d, \"q_ahead\": %d, \"ring\": %d, \"ar\": %d, \"pt\": %d, \"tpw\": %d, \"nw\": %d, \"gpw\": %d, \"hpw\": %d, \"halo\": %d, \"steps\": [", mode, s.ok,
         s.nsteps, s.npieces, s.ngroups, s.q_init, s.q_ahead, RING, AR, PT, TPW, NW, GPW, HPW, (int)kHalo);
  for (int i = 0; i < s.nsteps; ++i) {
    const StepD& d = s.st[i];
    printf("%s{\"in\": %d, \"chunk\": %d, \"g0\": %d, \"ng\": %d, \"dx0\": %d, \"ndx\": %d, \"tb\": %d, \"wp0\": %d, \"wpn\": %d, \"post\": %d, \"K\": %d, \"q0\": %d, "
           "\"q1\": %d, \"publish\": %d, \"tile_in\": %d, \"flag_in\": %d, \"Kflag\": %d, \"zero_mask\": %d, \"first_of_in\": %d, \"pre\": %d, \"nx_tile\": %d, "
           "\"nx_q0\": %d, \"nx_q1\": %d, \"claim\": %d, \"Kclaim\": %d, \"mask_conv\": %d, \"copy_in\": %d}",
           i ? ", " : "", d.in, d.chunk, d.g0, d.ng, d.dx0, d.ndx, d.tb, d.wp0, d.wpn, d.post, d.K, d.q0, d.q1, d.publish, d.tile_in, d.flag_in, d.Kflag,
           d.zero_mask, d.first_of_in, d.pre, d.nx_tile, d.nx_q0, d.nx_q1, d.claim, d.Kclaim, mask_conv_at(s, i), d.copy_in);
  }
  printf("]}\n");
}
int main() {
  for (int mode = 0; mode < 3; ++mode) dump(make_sched(mode), mode);
  return 0;
}
