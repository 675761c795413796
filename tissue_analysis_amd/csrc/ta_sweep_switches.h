// ta_sweep_switches.h -- every compile-time switch of the sweep (kernels_scan.hip, ta_sweep_common.h) in ONE place: what the product
// is built with (the defaults below), what experiments measured and left off, and the instrumentation / ablation builds of
// scripts/build_variant.py (`python scripts/build_variant.py NAME -DTA_...` -> scratch/libNAME.so; never the product library).
// The switches are USED where the code they change lives; this header only names and documents them.
#pragma once

// ---- tuning: what the product runs with ---------------------------------------------------------------------------------
#ifndef TA_WAVES
#define TA_WAVES 4               // waves per workgroup, stacked along axis 1
#endif
#ifndef TA_PSLOTS
#define TA_PSLOTS 512            // pair-table slots per workgroup
#endif
#ifndef TA_LSLOTS
#define TA_LSLOTS 128            // label-table slots per workgroup
#endif
#ifndef TA_FCAP
#define TA_FCAP 256              // face records a wave's buffer holds (the top-of-plane drain takes the LAST groups: nothing moves)
#endif
#ifndef TA_RCAP
#define TA_RCAP 128              // run records a wave's buffer holds (the drain takes the first 64, the rest moves to the front: <= 128)
#endif
#ifndef TA_FDRAIN
#define TA_FDRAIN 128            // faces in the buffer from which the top-of-plane drain takes two groups of 64
#endif
#ifndef TA_FDRAIN1
#define TA_FDRAIN1 1             // ... and one group from 64 on, in the tiles of eight voxels a lane (kernels_scan.hip, at its use)
#endif
#ifndef TA_DRAIN_ALL
#define TA_DRAIN_ALL 1           // the top-of-plane drains (full groups, every lookup in flight together)
#endif
#ifndef TA_DRAIN_ALL_U16
#define TA_DRAIN_ALL_U16 1       // ... for the full tiles of uint16 volumes too
#endif
#ifndef TA_XCD_CHUNK
#define TA_XCD_CHUNK 4           // consecutive tiles given to one XCD (0 = plain order).  Round 5: 8 .. 64 cut the HBM fetch by 2 .. 5 % and
#endif                           // cost 1 .. 12 % of time on C4 (tissue and background no longer mix over the XCDs); flat on C5
#ifndef TA_HOT_ADJ
#define TA_HOT_ADJ 1             // the hot (most common) label gets a private row per workgroup, also with adjacency
#endif
#ifndef TA_MASKED_STORES
#define TA_MASKED_STORES 7       // record stores predicated by the exec mask (v_cmpx); bits: 1 = a plane's axis-0 faces, 2 = a row's axis-1
#endif                           // faces, 4 = its runs; 0 = round 4's stores (own offset or trash slot by selects)
#ifndef TA_PROBE_NORTN
#define TA_PROBE_NORTN 1         // a probe round = a compare-and-swap that returns nothing + a plain re-read
#endif
#ifndef TA_PROLOGUE_OVERLAP
#define TA_PROLOGUE_OVERLAP 1    // the plane before a tile and the tile's first plane in flight together (tiles of eight voxels a lane)
#endif
#ifndef TA_FLUSH_BOX_READ
#define TA_FLUSH_BOX_READ 1      // the flush reads a label's global box and sends only the bounds its tile extends
#endif
#ifndef TA_FLUSH_TRANSPOSE
#define TA_FLUSH_TRANSPOSE 1     // the flush sends a label's sums lane = (label, word): consecutive lanes, consecutive words of a global row
#endif
#ifndef TA_PLANES_CAP_ADJ8
#define TA_PLANES_CAP_ADJ8 32    // the tallest tile the kernels of eight voxels a lane pack their sums for (SumPack)
#endif
#ifndef TA_U16_VPL
#define TA_U16_VPL 8             // uint16 volumes with adjacency: 8 voxels a lane (125 VGPRs, four waves) or 4 (the uint32 kernel's shape, five
#endif                           // waves: faster only where cells are everywhere -- profiles/NOTES.md, round 4 section 7)
#ifndef TA_U16_MOM_RB
#define TA_U16_MOM_RB 2          // rows a wave of the moments-only uint16 kernel (512 columns each)
#endif

// ---- built, measured, left OFF (kept because they are one flag away from a same-call A/B) -----------------------------------
#ifndef TA_LSUM_REP
#define TA_LSUM_REP 1            // 2: two replicas of a label slot's sums by row parity (halves the lanes sharing an add's address): +4 % time
#endif
#ifndef TA_PCNT64
#define TA_PCNT64 0              // 1: a pair's three face counts in one u64 LDS word (21 bits each): equal time, twice the cost per atomic
#endif
#ifndef TA_PERSIST
#define TA_PERSIST 0             // 1: persistent workgroups on per-XCD tile queues for the narrow uint32 kernel: 1.30 against 1.07 ms (round 4)
#endif
#ifndef TA_PERSIST_WGS
#define TA_PERSIST_WGS (256 * 5) // ... how many
#endif
#ifndef TA_PERSIST_SINGLE_QUEUE
#define TA_PERSIST_SINGLE_QUEUE 0
#endif
#ifndef TA_PLANES_IN_FLIGHT
#define TA_PLANES_IN_FLIGHT 1    // 2: a second landing zone for the narrow uint32 kernel (109 VGPRs, four waves): slower (round 4)
#endif

// ---- ablations (results WRONG by construction; only the time matters) and instrumentation ---------------------------------------
//   TA_ABLATE = 1 records produced and stored, nothing consumed; 2 = placed, not stored; 3 = not even placed
//   TA_ABL_HOT = 1 the top-of-plane drains add nothing to the tables; 2 ... and run no probe rounds; 3 ... and read no keys; 4 ... nor records
//   TA_ABL_NOFLUSH the tile tables are not flushed;  TA_ABL_NOFLUSH_PAIRS / _LABELS only one kind is
//   TA_ABL_NOSLOW the in-plane drains and the end of the tile consume nothing;  TA_ABL_NOHOT no top-of-plane drains
//   TA_ABL_NOFACE0 / TA_ABL_NOFACE1 no axis-0 / axis-1 face records;  TA_ABL_NOSUMS, TA_ABL_NOBOX, TA_ABL_NOBOXHOT, TA_ABL_NOPCNT, TA_ABL_NOLOOP,
//   TA_ABL_SHARE1 (every lane of an add its own address), TA_ABL_L2 (every plane re-reads the tile's first: an L2-resident run)
//   TA_ABL_NOHALO = 1 a workgroup's first wave reads no row above (the row another tile owns: 12.5 % of the bytes fetched); 2 = no wave does
//   TA_FLUSH_SCOPE the memory scope of the flush's global adds (gfx950 encodes agent and workgroup scope alike: nothing to measure)
//   TA_RECCOUNT   flags[8..12] = face / run records and calls through the in-plane drain, faces / runs through the top-of-plane drains
//   TA_BARSTAMP   flags[8..11] = cycles >> 8 over the waves: sweep of the tile, wait at the barrier before the flush, flush; wave-tiles
//   TA_STAMPS     per-phase s_memtime stamps of the narrow kernels (scripts/probe_stamps.py);  TA_DBG_EMIT an in-kernel check of the
//                 predicated stores' offsets;  TA_LDS_PAD extra LDS per workgroup (fewer workgroups a CU)
#ifndef TA_ABLATE
#define TA_ABLATE 0
#endif
#ifndef TA_ABL_HOT
#define TA_ABL_HOT 0
#endif
