// Compile-time variants of the kernels that are kept on purpose (everything else that was tried is recorded in DESIGN.md and
// removed from the sources).  tests/test_abi_and_host.py::test_kept_variants_compile builds each of them for gfx950.
#pragma once

// EGO_GATHER_TEAMS (ego_shade.hip, default 1): 1 = the appearance gather of the fp16-split shade kernels runs in 4-lane teams that
// read whole 64-byte lines and exchange quads through DPP; 0 = every lane gathers its own sample's quads, no lane talks to another.
// Kept because it separates the reproducibility fault of DESIGN.md 5.1 from the lane exchange (the fault also hits this form).
#ifndef EGO_GATHER_TEAMS
#define EGO_GATHER_TEAMS 1
#endif

// EGO_PAIRED_WEIGHTS (ego_shade.hip, default undefined): defined = do NOT pin the four bilinear weights to registers of their own.
// With the SLP vectoriser on (build with EGO_NO_PER_FILE_FLAGS=1) the compiler then forms {w00, w01} pairs and emits packed fp32
// instructions that broadcast the HIGH half of a pair - the one code-generation feature every non-reproducible build had in common
// (DESIGN.md 5.1).  THIS IS THE KNOWN-FAULTY FORM: it exists only as the reproducer of that fault, never ship it.

// EGO_HOIST_FOLD / EGO_HOIST_PLAIN (ego_shade.hip): how many of a plane's three basis fragment pairs the rolling gather loads BEFORE it
// issues the next plane's tap loads (vmcnt retires in order: a fragment loaded behind the taps makes its MFMA wait for all of them).
// 3 in the folded kernel (k_shade_h<.., FOLD = true>: no spill, -3.4 % kernel time); the two-launch kernel has no registers for it
// (3 -> 32 VGPRs spilled inside the tile loop, +9 %).
#ifndef EGO_HOIST_FOLD
#define EGO_HOIST_FOLD 3
#endif
#ifndef EGO_HOIST_PLAIN
#define EGO_HOIST_PLAIN 0
#endif

// EGO_WALK_PROF (ego_scatter_sorted.hip, default undefined): an INSTRUMENT, not a variant of the arithmetic - the walk kernel of the
// sorted scatter reads s_memtime around its phases (step head + set-up, load issue, wait, compute, tail) and sums the cycles of all
// waves into device counters, per wave the busy time, steps and iterations (ego_debug_walk_prof / ego_debug_walk_waves, read by
// tools/sorted_probe.py with PROBE_PROF=1 on a tools/build_variant.sh build).  It is how round 6 found that the first walk's slowest
// wave took 3.5 x the mean (chunks of cells dealt by count) and that the 48-channel walk is bound by VALU issue, not by its gathers.

// EGO_GENERIC_MFMA (ego_generic.hip, default 1): 1 = the any-shape head (basis, both hidden layers, the backward chain) runs on the matrix
// pipe in fp32 (v_mfma_f32_32x32x2_f32 over slab-staged inputs); 0 = the lane = sample form of rounds 2-5 (every FMA with an SGPR weight
// operand, one wave per SIMD: 5-10 % of the fp32 rate) - kept as the plain restatement of the reference's op order to hold the MFMA form
// against (NOTEBOOK 10.7: inference 18 -> 2.4 ms, training 50 -> 15.5 ms on the MLP-head shape).
