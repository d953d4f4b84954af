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
