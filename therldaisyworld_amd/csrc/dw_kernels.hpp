// dw_kernels.hpp — HIP kernels of the RLDaisyWorld hot path for gfx950 (MI355X), one header per family:
//
//   dw_physics.hpp        per-cell arithmetic (float64 staging; fused float32 algebra, packed two-cell form)
//   dw_common.hpp         reductions, adaptors, the four-cell row group `cells4`
//   dw_step_generic.hpp   step_generic      one thread per cell, any shape / float64: the in-library reference
//   dw_step_tiled.hpp     step_tiled        LDS tile + halo (W < 256), global near-tie queues, fix-up kernels
//   dw_step_stream.hpp    step_stream_*     wave-strip streaming kernel (W >= 256): the single-step hot kernel
//   dw_step_fused.hpp     step_stream_fused2[_exact]   two steps per HBM round trip (dw_step_n): the headline
//   dw_step_first.hpp     step_first_stream  the first step of an episode (un-quantised input), W a multiple of 256
//   dw_episode.hpp        episode_small     K steps in one launch with the worlds in LDS (H*W <= 4096)
//   dw_episode_wave.hpp   episode_wave      the same for H*W <= 256: one wave per world, no workgroup barrier in the step
//   dw_agents.hpp         agents_update (ref :181-244), observe (ref get_obs :246-263), policy_greedy
//                         (agents/greedy.py:14-36), policy_mlp (agents/mlp.py:97-116), reward/done, lifespans
//   dw_agents_fused.hpp   agents_lookahead_patch: the agents' step between the two steps of a fused launch
//   dw_state_io.hpp       materialise (ref self.grid :445-459 / :304-323), init_random (Philox), conversions
//
// Wavefront = 64 lanes, 256-thread workgroups (4 waves), no MFMA.  Planes are binary16 per-mille integers: 8
// algorithmic bytes per cell-update (2 planes read + 2 written).  Measured bounds (DESIGN.md sections 3 and 6): the
// fused step pairs AND the single-step wave-strip kernels issue VALU work >= 85-95 % of the time (35 / 51 and
// 39 / 56 instructions per cell-evaluation, float32-only / exact); HBM runs at 0.4-0.6 of its peak.
#pragma once
#include "dw_common.hpp"
#include "dw_step_generic.hpp"
#include "dw_step_tiled.hpp"
#include "dw_step_stream.hpp"
#include "dw_step_fused.hpp"
#include "dw_step_first.hpp"
#include "dw_episode.hpp"
#include "dw_episode_wave.hpp"
#include "dw_agents.hpp"
#include "dw_state_io.hpp"
#include "dw_agents_fused.hpp"
