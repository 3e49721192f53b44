// The whole single-rank update as ONE launch (option "one_launch"; DESIGN.md 4.5: bit-identical to the two launches,
// measured slower, kept as a tested alternative).  Needs mlp_rows.h (the row groups) and mlp_lean_gemm.h (the tiles).
#pragma once

// The whole single-rank update in ONE launch of B workgroups (curious_ddpg_update on the row-local route): the row
// groups of ddpg_rows_her_kernel (the spare quarter gathers the NEXT batch); a workgroup that is through with its rows
// then works as tile worker `w` (its role index) on the tiles of dw_adam_her_kernel -- weight / bias gradients with Adam
// in the epilogue, the loss block -- from two lists:
//   C  the critic's tiles (hidden matrices, then the compact small list): tile i goes to worker i mod 3B/4 of the spare,
//      target and main-critic workgroups; they run while the actor-side chain is in its last three layers
//   A  the actor's tiles and the loss block: tile i goes to worker i mod B of all workgroups; they run when that chain ends
// A worker prefetches theta / m / v of its tile, then waits (mlp_common.h step_wait) for the row groups that produce
// what the tile reads.  No launch boundary between the two halves of an update (~1.3 us idle, the ramp of a fresh grid,
// the cold start of every tile at once), and nothing depends on dispatch order: every workgroup of the launch is
// resident from the start (B workgroups <= CUs is required) and first does its rows, then waits.
// Same arithmetic as the two launches, bit for bit.
struct StepPlan { int32_t hot_c, small_c, hot_a, small_a; };   // tiles: critic hidden / small, actor hidden / small
__global__ __launch_bounds__(256) void ddpg_step_kernel(RowsArgs a, Ex ex, DwAllArgs args, AdamFuse A, HerArgs her,
                                                        StepPlan plan) {
  extern __shared__ __attribute__((aligned(16))) float rows_lds[];
  const int wk = ddpg_rows_body<false, true, true>(a, ex, &her, 0);
  const int nrg = a.B / ROWS_R;
  StepSync S;
  // (a tile has to outlast the consumers of Q' it waits behind: 8 x their patience)
  S.c = a.sync; S.n = nrg; S.fault = a.fault; S.lab = a.lab_step;
  S.spins = (a.spins > 0 && a.spins < (1 << 27)) ? 8 * a.spins : (1 << 30);
  S.st = nullptr;
  int nst = 0;
  const int nC = plan.hot_c + plan.small_c, nA = plan.hot_a + plan.small_a + 1;
  if (wk < 3 * nrg) {
    for (int i = wk; i < nC; i += 3 * nrg) {
      __syncthreads();                                       // (LDS and step_wait's flag are reused from tile to tile)
      if (a.stamps && wk % nrg == 0 && nst < 4) {            // lab: workers 0, nrg, 2 nrg
        S.st = a.stamps + 128 + 8 * (4 * (wk / nrg) + nst++);
        if (threadIdx.x == 0) S.st[0] = __builtin_readcyclecounter();
      }
      if (i < plan.hot_c) dw_hot_body<true>(args.hot, A, i, rows_lds, 0, 0, &S);
      else dw_small_item<true>(args.small, A, i - plan.hot_c, rows_lds, 0, 0, &S);
      if (S.st && threadIdx.x == 0) S.st[4] = __builtin_readcyclecounter();
      S.st = nullptr;
    }
  }
  for (int i = wk; i < nA; i += 4 * nrg) {
    __syncthreads();
    if (a.stamps && wk % nrg == 0 && nst < 4) {
      S.st = a.stamps + 128 + 8 * (4 * (wk / nrg) + nst++);
      if (threadIdx.x == 0) S.st[0] = __builtin_readcyclecounter();
    }
    if (i < plan.hot_a) dw_hot_body<true>(args.hot, A, plan.hot_c + i, rows_lds, 0, 0, &S);
    else if (i < nA - 1) dw_small_item<true>(args.small, A, plan.small_c + (i - plan.hot_a), rows_lds, 0, 0, &S);
    else dw_loss_fin(args.small.fin, rows_lds, 0, 0, &S);
    if (S.st && threadIdx.x == 0) S.st[4] = __builtin_readcyclecounter();
    S.st = nullptr;
  }
  step_ticket(a.sync, a.n_tickets, a.step_ctr);
}

