"""Generate tissue_analysis_amd/csrc/ta_pin_tables.inc: the Pin<BASE> specialisations of kernels_scan.hip.

    python scripts/gen_pin_tables.py            # rewrite the .inc
    python scripts/gen_pin_tables.py --check    # exit 1 when the committed .inc differs (tests/test_capi_symbols.py runs this)

A Pin<BASE> names, by number, the VGPRs above a kernel's amdgpu_num_vgpr budget in which the plane in flight lands
(register numbers must be literals inside inline asm, so the tables are written out by this script rather than by hand):
  with adjacency (21 registers): rows 0..3 of the wave tile = v[BASE .. BASE+15], the row above = v[BASE+16 .. BASE+19], the
    voxels left of the rows = v[BASE+20]; with two rows per wave (RB = 2) the row above is the third quad and the voxel v[BASE+12];
  without adjacency (16 registers): the rows only.
The budgets (TA_PIN_*) are the #defines next to the #include in kernels_scan.hip."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tissue_analysis_amd", "csrc", "ta_pin_tables.inc")
TABLES = [(104, True), (120, True), (76, False), (80, False)]      # (first pinned register, with adjacency)
TWO_ROW_TABLES = [82, 112, 96]  # adjacency, TWO rows per wave only: 13 registers (rows, row above, voxel to the left); 112: the padded tiles;
                                # 95: the second landing zone of the two-planes-in-flight experiment (behind 82's)


WIDE_TABLES = [102, 116]        # adjacency, two rows of eight uint32 voxels a lane (25 registers); 116: the padded tiles


def clobbers(base, n):
    return '"memory", ' + ", ".join('"v%d"' % (base + i) for i in range(n))


def movs(base, n):
    return " ".join('"v_mov_b32 %%%d, v%d\\n"' % (i, base + i) for i in range(n))


def half_strips(base, quads, c):
    """issue_half<Q>: an 8-byte strip (four voxels of a uint16 volume) into the first two registers of quad Q."""
    o = ["    template <int Q> static __device__ __forceinline__ void issue_half(uint32_t voff, const void* sbase) {"]
    for q in range(quads):
        head = "if (Q == %d)" % q if q == 0 else ("else if (Q == %d)" % q if q < quads - 1 else "else")
        o.append('        %s asm volatile("global_load_dwordx2 v[%d:%d], %%0, %%1" :: "v"(voff), "s"(sbase) : %s);'
                 % (head, base + 4 * q, base + 4 * q + 1, c))
    o.append("    }")
    return o


def table(base, adj):
    n = 21 if adj else 16
    c = clobbers(base, n)
    quads = 5 if adj else 4
    o = ["template <> struct Pin<%d> {" % base,
         "    template <int Q> static __device__ __forceinline__ void issue_strip(uint32_t voff, const void* sbase) {"]
    for q in range(quads):
        head = "if (Q == %d)" % q if q == 0 else ("else if (Q == %d)" % q if q < quads - 1 else "else")
        o.append('        %s asm volatile("global_load_dwordx4 v[%d:%d], %%0, %%1" :: "v"(voff), "s"(sbase) : %s);'
                 % (head, base + 4 * q, base + 4 * q + 3, c))
    o.append("    }")
    o += half_strips(base, quads, c)
    if adj:
        o.append("    template <typename T, int RB> static __device__ __forceinline__ void issue_voxel(uint32_t voff, const void* sbase) {")
        for wide, four, reg in ((True, True, base + 20), (True, False, base + 12), (False, True, base + 20), (False, False, base + 12)):
            o.append('        if (sizeof(T) %s 4 && RB %s 4) asm volatile("global_load_%s v%d, %%0, %%1" :: "v"(voff), "s"(sbase) : %s);'
                     % ("==" if wide else "!=", "==" if four else "!=", "dword" if wide else "ushort", reg, c))
        o.append("    }")
    else:
        o.append("    template <typename T, int RB> static __device__ __forceinline__ void issue_voxel(uint32_t, const void*) {}")
    o.append("    template <int RB> static __device__ __forceinline__ void landed(u32x4 (&raw)[RB], u32x4& upr, uint32_t& l) {")
    rows4 = ('"=&v"(raw[0].x), "=&v"(raw[0].y), "=&v"(raw[0].z), "=&v"(raw[0].w),\n'
             '                           "=&v"(raw[1].x), "=&v"(raw[1].y), "=&v"(raw[1].z), "=&v"(raw[1].w),\n'
             '                           "=&v"(raw[RB > 2 ? 2 : 0].x), "=&v"(raw[RB > 2 ? 2 : 0].y), "=&v"(raw[RB > 2 ? 2 : 0].z), "=&v"(raw[RB > 2 ? 2 : 0].w),\n'
             '                           "=&v"(raw[RB > 3 ? 3 : 0].x), "=&v"(raw[RB > 3 ? 3 : 0].y), "=&v"(raw[RB > 3 ? 3 : 0].z), "=&v"(raw[RB > 3 ? 3 : 0].w)')
    rows2 = ('"=&v"(raw[0].x), "=&v"(raw[0].y), "=&v"(raw[0].z), "=&v"(raw[0].w),\n'
             '                           "=&v"(raw[1].x), "=&v"(raw[1].y), "=&v"(raw[1].z), "=&v"(raw[1].w)')
    halo = ', "=&v"(upr.x), "=&v"(upr.y), "=&v"(upr.z), "=&v"(upr.w), "=&v"(l)' if adj else ""
    o.append("        if (RB == 4) {")
    o.append('            asm volatile("s_waitcnt vmcnt(0)\\n" %s' % movs(base, 21 if adj else 16))
    o.append("                         : %s%s" % (rows4, halo))
    o.append('                         :: "memory");')
    o.append("        } else {")
    o.append('            asm volatile("s_waitcnt vmcnt(0)\\n" %s' % movs(base, 13 if adj else 8))
    o.append("                         : %s%s" % (rows2, halo))
    o.append('                         :: "memory");')
    o.append("        }")
    if not adj:
        o.append("        (void)upr; (void)l;")
    o.append("    }")
    o.append("};")
    return "\n".join(o)


def table_two_rows(base):
    c = clobbers(base, 13)
    o = ["template <> struct Pin<%d> {" % base,
         "    template <int Q> static __device__ __forceinline__ void issue_strip(uint32_t voff, const void* sbase) {",
         "        // (Q > 2 is never issued with two rows per wave; the branch exists only because the caller's `if (RB > 2)` is not constexpr)"]
    for q in range(3):
        head = "if (Q == %d)" % q if q == 0 else ("else if (Q == %d)" % q if q < 2 else "else")
        o.append('        %s asm volatile("global_load_dwordx4 v[%d:%d], %%0, %%1" :: "v"(voff), "s"(sbase) : %s);'
                 % (head, base + 4 * q, base + 4 * q + 3, c))
    o.append("    }")
    o += half_strips(base, 3, c)
    o.append("    template <typename T, int RB> static __device__ __forceinline__ void issue_voxel(uint32_t voff, const void* sbase) {")
    o.append('        static_assert(RB == 2, "this budget holds two rows");')
    o.append('        if (sizeof(T) == 4) asm volatile("global_load_dword v%d, %%0, %%1" :: "v"(voff), "s"(sbase) : %s);' % (base + 12, c))
    o.append('        else asm volatile("global_load_ushort v%d, %%0, %%1" :: "v"(voff), "s"(sbase) : %s);' % (base + 12, c))
    o.append("    }")
    o.append("    template <int RB> static __device__ __forceinline__ void landed(u32x4 (&raw)[RB], u32x4& upr, uint32_t& l) {")
    o.append('        static_assert(RB == 2, "this budget holds two rows");')
    o.append('        asm volatile("s_waitcnt vmcnt(0)\\n" %s' % movs(base, 13))
    o.append('                     : "=&v"(raw[0].x), "=&v"(raw[0].y), "=&v"(raw[0].z), "=&v"(raw[0].w),')
    o.append('                       "=&v"(raw[1].x), "=&v"(raw[1].y), "=&v"(raw[1].z), "=&v"(raw[1].w), "=&v"(upr.x), "=&v"(upr.y), "=&v"(upr.z), "=&v"(upr.w), "=&v"(l)')
    o.append('                     :: "memory");')
    o.append("    }")
    o.append("    // (the same with the FOUR youngest loads -- the next plane's, in the other landing zone -- left in flight)")
    o.append("    template <int RB> static __device__ __forceinline__ void landed_keep4(u32x4 (&raw)[RB], u32x4& upr, uint32_t& l) {")
    o.append('        static_assert(RB == 2, "this budget holds two rows");')
    o.append('        asm volatile("s_waitcnt vmcnt(4)\\n" %s' % movs(base, 13))
    o.append('                     : "=&v"(raw[0].x), "=&v"(raw[0].y), "=&v"(raw[0].z), "=&v"(raw[0].w),')
    o.append('                       "=&v"(raw[1].x), "=&v"(raw[1].y), "=&v"(raw[1].z), "=&v"(raw[1].w), "=&v"(upr.x), "=&v"(upr.y), "=&v"(upr.z), "=&v"(upr.w), "=&v"(l)')
    o.append('                     :: "memory");')
    o.append("    }")
    o.append("};")
    return "\n".join(o)


def table_wide(base):
    """Two rows of EIGHT uint32 voxels a lane: six quads (row 0 low / high half, row 1 low / high, the row above low / high)
    and the voxel to the left: 25 registers."""
    c = clobbers(base, 25)
    o = ["template <> struct Pin<%d> {" % base,
         "    template <int Q> static __device__ __forceinline__ void issue_strip(uint32_t voff, const void* sbase) {"]
    for q in range(6):
        head = "if (Q == %d)" % q if q == 0 else ("else if (Q == %d)" % q if q < 5 else "else")
        o.append('        %s asm volatile("global_load_dwordx4 v[%d:%d], %%0, %%1" :: "v"(voff), "s"(sbase) : %s);'
                 % (head, base + 4 * q, base + 4 * q + 3, c))
    o.append("    }")
    o.append("    template <typename T, int RB> static __device__ __forceinline__ void issue_voxel(uint32_t voff, const void* sbase) {")
    o.append('        static_assert(sizeof(T) == 4 && RB == 2, "this layout holds two rows of a uint32 volume");')
    o.append('        asm volatile("global_load_dword v%d, %%0, %%1" :: "v"(voff), "s"(sbase) : %s);' % (base + 24, c))
    o.append("    }")
    o.append("    static __device__ __forceinline__ void landed_wide(u32x4 (&q)[6], uint32_t& l) {")
    o.append('        asm volatile("s_waitcnt vmcnt(0)\\n" %s' % movs(base, 25))
    outs = ["\"=&v\"(q[%d].x), \"=&v\"(q[%d].y), \"=&v\"(q[%d].z), \"=&v\"(q[%d].w)" % (k, k, k, k) for k in range(6)]
    o.append("                     : " + (",\n                       ").join(outs) + ', "=&v"(l)')
    o.append('                     :: "memory");')
    o.append("    }")
    o.append("};")
    return "\n".join(o)


def render():
    head = ("// ta_pin_tables.inc -- GENERATED by scripts/gen_pin_tables.py (do not edit; `--check` compares): the hand-pinned landing\n"
            "// registers of the sweep kernels, one Pin<BASE> per VGPR budget, BASE = the first register above amdgpu_num_vgpr.\n")
    return head + "\n".join([table(b, a) for b, a in TABLES] + [table_two_rows(b) for b in TWO_ROW_TABLES]
                            + [table_wide(b) for b in WIDE_TABLES]) + "\n"


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print(OUT)
