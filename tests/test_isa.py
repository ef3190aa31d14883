"""Static guards over the gfx950 listings the compiler actually emitted (CPU test: hipcc cross-compiles, no GPU needed).

Round 2 found, by luck of a fuzz draw, a DPP hazard inside a generated inline-asm chunk: hipcc does not look into asm
statements, so nothing inserts the wait states there, and a toolchain bump can re-create such a placement silently.  These
tests read `hipcc -S` listings built with the library's own flags (accelerated-tinympc_amd/build.py: device_asm) and check

  1. every DPP instruction: no VALU instruction in the two wait states in front of it writes the VGPR it reads through the DPP
     path (gfx950: VALU write VGPR -> DPP read of that VGPR needs 2 wait states);
  2. no VALU write of EXEC (v_cmpx*, v_readlane-style writes do not count) within 5 wait states in front of a DPP instruction;
  3. scratch (spill) sizes of the headline kernels stay at their pinned values;
  4. v_pk_add_f32 appears in no exact-arithmetic kernel built with -fno-slp-vectorize except admm_tile16.hip, whose 4-vector sums
     are deliberate (one wave per SIMD: a packed add is two separately rounded IEEE adds at one issue slot).

The checker itself is tested on a listing with a chunk's leading `s_nop 1` removed."""
import re

import pytest

import accelerated_tinympc_amd as T

DPP_SOURCES = ["admm_rowlane.hip", "admm_rowloop.hip", "admm_quadlane.hip", "admm_steps.hip", "dispatch_order.hip"]
VALU_EXEC_WRITERS = ("v_cmpx",)


def kernels_of(listing: str):
    """{mangled name: [instruction lines]} for every kernel (function body up to its .amdhsa_kernel / end marker)."""
    out, cur, name = {}, None, None
    for line in listing.split("\n"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end") or line.lstrip().startswith(".amdhsa_kernel") or line.lstrip().startswith(".section"):
            cur = None
            continue
        t = line.strip()
        if not t or t[0] in ";." and not re.match(r"^\.LBB\d+_\d+:", t):
            continue
        cur.append(t.split(";")[0].strip())
    return out


def regs(op: str):
    """VGPR numbers named by one operand: v7 -> {7}, v[4:7] -> {4,5,6,7}; anything else -> {}."""
    op = op.strip().rstrip(",")
    op = re.sub(r"^[-|]+|[|]+$", "", op)
    m = re.match(r"^v(\d+)$", op)
    if m:
        return {int(m.group(1))}
    m = re.match(r"^v\[(\d+):(\d+)\]$", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def is_dpp(ins: str):
    return "_dpp" in ins.split()[0] or re.search(r"\b(row_newbcast|row_shr|row_shl|row_ror|row_bcast|quad_perm|row_mirror|row_half_mirror|wave_shr|wave_ror|row_share|row_xmask):?", ins) is not None


def hazards(lines):
    """[(index, text, reason)] of DPP instructions with a VALU write of their DPP source less than 2 wait states ahead, or a VALU
    write of EXEC less than 5 wait states ahead.  Labels end the look-back (another path may join there)."""
    bad = []
    for i, ins in enumerate(lines):
        if ins.endswith(":") or not ins.startswith("v_") or not is_dpp(ins):
            continue
        ops = [o for o in re.split(r",\s*", ins.split(None, 1)[1])] if len(ins.split(None, 1)) > 1 else []
        if len(ops) < 2:
            continue
        src = regs(ops[1].split()[0])  # src0 is the operand that goes through the DPP network
        ws, j = 0, i - 1
        while j >= 0 and ws < 5:
            p = lines[j]
            if p.endswith(":"):
                break
            op = p.split()[0]
            if op == "s_nop":
                ws += int(p.split()[1]) + 1
                j -= 1
                continue
            if op.startswith("v_") and not op.startswith(("v_cmp_", "v_readlane", "v_readfirstlane")):
                if op.startswith(VALU_EXEC_WRITERS):
                    bad.append((i, ins, f"VALU write of EXEC {ws} wait states ahead: {p}"))
                elif ws < 2:
                    pops = p.split(None, 1)[1] if len(p.split(None, 1)) > 1 else ""
                    dst = regs(re.split(r",\s*", pops)[0]) if pops else set()
                    if dst & src:
                        bad.append((i, ins, f"VALU write of the DPP source {ws} wait states ahead: {p}"))
            ws += 1
            j -= 1
    return bad


@pytest.fixture(scope="module")
def listings():
    return {src: T.build.device_asm(src).read_text() for src in DPP_SOURCES + ["admm_tile16.hip", "admm_tile16_pi.hip", "admm_tile48.hip", "admm_waveres.hip", "admm_wave.hip"]}


def test_the_checker_flags_a_removed_wait_state(listings):
    """Self-test: take a kernel of the real listing, drop the `s_nop 1` in front of a DPP group whose source was written by the
    VALU instruction just before, and the checker must report it (while the unmodified kernel is clean)."""
    flagged = 0
    for name, lines in kernels_of(listings["admm_rowlane.hip"]).items():
        if "admm_rowlane_kernel" not in name:
            continue
        assert not hazards(lines), name
        for i in range(2, len(lines)):
            if lines[i - 1].startswith("s_nop 1") and is_dpp(lines[i]) and lines[i - 2].startswith("v_") and not lines[i - 2].endswith(":"):
                ops = re.split(r",\s*", lines[i].split(None, 1)[1])
                dst_prev = regs(re.split(r",\s*", lines[i - 2].split(None, 1)[1])[0])
                if dst_prev & regs(ops[1].split()[0]):
                    broken = lines[:i - 1] + lines[i:]
                    assert hazards(broken), ("the checker missed a removed s_nop in front of", lines[i], "after", lines[i - 2])
                    flagged += 1
                    break
        if flagged:
            break
    assert flagged, "no DPP group directly behind a VALU write of its source found to test the checker on"


@pytest.mark.parametrize("src", DPP_SOURCES)
def test_no_dpp_hazard_in_any_kernel(listings, src):
    for name, lines in kernels_of(listings[src]).items():
        bad = hazards(lines)
        assert not bad, (src, name, bad[:3])


# bytes of scratch per lane of the headline instantiations as built today: a compiler or source change that makes one of them
# spill (more) must be looked at, not discovered as a slow-down
SCRATCH_PINS = {
    # <NX, NU, N, EXACT, H16, MPC, BPI, D32, OPT>
    ("admm_rowlane.hip", "admm_rowlane_kernelILi12ELi4ELi30ELb1ELb0ELb0ELb0ELb0ELb0EEE"): 68,   # exact: the 17 registers of DESIGN.md 5.1
    ("admm_rowlane.hip", "admm_rowlane_kernelILi12ELi4ELi30ELb0ELb0ELb0ELb0ELb0ELb0EEE"): 0,    # fma
    ("admm_rowlane.hip", "admm_rowlane_kernelILi12ELi4ELi30ELb1ELb0ELb0ELb0ELb0ELb1EEE"): 0,    # exact with the optional terms (round 4)
    # the persistent tile16 kernels keep their tile-invariant values (gains, tables' bases, the -0 accumulator) live across the whole
    # tile body and spill a few dozen registers around prologue and epilogue; the ITERATION LOOP must stay (almost) free of scratch
    # traffic, which test_tile16_iteration_loop_is_free_of_scratch_traffic checks separately
    # <N, EXACT, COLD, MPC>; MPC = the closed loop on chip (round 4)
    # <N, EXACT, COLD, MPC, BR, XR>
    ("admm_tile16.hip", "admm_tile16_kernelILi30ELb1ELb1ELb0ELb0ELb0EEE"): 32,   # (28 before the two-ended tile queue: one more scalar across the tile loop)
    ("admm_tile16.hip", "admm_tile16_kernelILi30ELb1ELb0ELb0ELb0ELb0EEE"): 188,
    ("admm_tile16.hip", "admm_tile16_kernelILi30ELb0ELb1ELb0ELb0ELb0EEE"): 0,
    ("admm_tile16.hip", "admm_tile16_kernelILi30ELb0ELb0ELb0ELb0ELb0EEE"): 0,
    # (the MPC instantiations spill around the block between two solves — plant step, deferred sweep, slack restore — which runs once per
    #  MPC step; their iteration loop is held to the same bound as the others by test_tile16_iteration_loop_is_free_of_scratch_traffic)
    ("admm_tile16.hip", "admm_tile16_kernelILi30ELb1ELb1ELb1ELb0ELb0EEE"): 280,
    ("admm_tile16.hip", "admm_tile16_kernelILi30ELb1ELb0ELb1ELb0ELb0EEE"): 268,
    ("admm_tile16.hip", "admm_tile16_kernelILi30ELb0ELb1ELb1ELb0ELb0EEE"): 0,
    ("admm_tile16.hip", "admm_tile16_kernelILi30ELb0ELb0ELb1ELb0ELb0EEE"): 0,
    # the "pi" instantiations (BR: bounds, XR: reference through per-wave LDS-DMA slots; admm_tile16_pi.hip), exact arithmetic, cold / warm start
    ("admm_tile16_pi.hip", "admm_tile16_kernelILi30ELb1ELb1ELb0ELb1ELb1EEE"): 88,
    ("admm_tile16_pi.hip", "admm_tile16_kernelILi30ELb1ELb0ELb0ELb1ELb1EEE"): 200,
    ("admm_tile16_pi.hip", "admm_tile16_kernelILi30ELb1ELb1ELb0ELb1ELb0EEE"): 48,
    ("admm_tile16_pi.hip", "admm_tile16_kernelILi30ELb1ELb0ELb0ELb1ELb0EEE"): 188,
    ("admm_tile16_pi.hip", "admm_tile16_kernelILi30ELb1ELb1ELb0ELb0ELb1EEE"): 48,
    ("admm_tile16_pi.hip", "admm_tile16_kernelILi30ELb1ELb0ELb0ELb0ELb1EEE"): 192,
    # the nx = 32 tile kernel keeps no state in registers (duals in LDS, slack streamed): nothing may spill
    # (round 4: <EXACT, TWO> — TWO is the instantiation for horizons whose duals leave room for a second workgroup per CU: it must also stay
    #  within 256 registers, which test_tile48_short_horizon_instantiation_fits_two_waves_per_simd checks)
    ("admm_tile48.hip", "admm_tile48_kernelILb1ELb0EEE"): 0,
    ("admm_tile48.hip", "admm_tile48_kernelILb0ELb0EEE"): 0,
    ("admm_tile48.hip", "admm_tile48_kernelILb1ELb1EEE"): 0,
    ("admm_tile48.hip", "admm_tile48_kernelILb0ELb1EEE"): 0,
    ("admm_waveres.hip", "admm_waveres_kernelILi32ELi16ELb1EEE"): 0,   # gains loaded per sweep since round 3
    ("admm_waveres.hip", "admm_waveres_kernelILi32ELi16ELb0EEE"): 0,
}


def scratch_sizes(listing: str):
    out, name = {}, None
    for line in listing.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
        m = re.match(r"^; ScratchSize: (\d+)", line)
        if m and name:
            out[name] = int(m.group(1))
    return out


def test_scratch_sizes_of_the_headline_kernels_are_pinned(listings):
    for (src, key), pin in SCRATCH_PINS.items():
        sizes = {k: v for k, v in scratch_sizes(listings[src]).items() if key in k}
        assert len(sizes) == 1, (src, key, list(sizes))
        (name, got), = sizes.items()
        assert got <= pin, f"{name}: {got} bytes of scratch per lane, pinned at {pin}"


def test_tile48_short_horizon_instantiation_fits_two_waves_per_simd(listings):
    """admm_tile48_kernel<EXACT, TWO = true> serves N <= 24, where two workgroups share a CU (LDS 80 KB each): that only happens while the kernel
    needs at most 256 registers (round 4: a uniform branch around one store had pushed the exact instantiation to 266 and cost 17 %)."""
    txt = listings["admm_tile48.hip"]
    for key in ("admm_tile48_kernelILb1ELb1EEE", "admm_tile48_kernelILb0ELb1EEE"):
        i = re.search(r"^_Z\w*" + key + r"\w*:", txt, re.M).start()   # the kernel's label; its resource summary follows the body
        m = re.search(r"; TotalNumVgprs: (\d+)", txt[i:])
        assert m and int(m.group(1)) <= 256, (key, m and m.group(1))


def test_packed_adds_only_where_they_are_deliberate(listings):
    """-fno-slp-vectorize keeps hipcc from pairing the scalar adds of the 16-lane and wave kernels into v_pk_add_f32 (slower at two
    waves per SIMD); admm_tile16.hip writes its sums over 4-vectors on purpose and must contain them."""
    for src in DPP_SOURCES + ["admm_waveres.hip", "admm_wave.hip"]:
        assert "v_pk_add_f32" not in listings[src], src
    t16 = kernels_of(listings["admm_tile16.hip"])
    exact = [l for n, l in t16.items() if "admm_tile16_kernelILi30ELb1E" in n]
    assert exact and all(sum(1 for i in l if i.startswith("v_pk_add_f32")) > 500 for l in exact)


def iteration_loop(lines):
    """(first, last) line of the ADMM iteration loop of a tile16 kernel: the innermost loop of the listing that contains the MFMAs of a whole
    iteration (58 unrolled sweep steps); the tile-queue loop around it is longer."""
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), 10 ** 9) < i:
            loops.append((labels[m.group(1)], i))
    with_mfma = [(a, b) for a, b in loops if sum(1 for l in lines[a:b] if l.startswith("v_mfma")) >= 200]
    assert with_mfma
    return min(with_mfma, key=lambda t: t[1] - t[0])


def test_tile16_iteration_loop_is_free_of_scratch_traffic(listings):
    """Inside the ADMM iteration loop of the headline kernel (the innermost loop of the listing that contains MFMAs: 58 unrolled sweep
    steps) a few scratch accesses per ITERATION are tolerated (today 0 in the cold-start and 31 in the warm-start instantiation, 1 in the closed-loop ones; 7 / 33 before round 4 took the Q registers and the bounds-table addresses out of the loop: -3.4 % kernel time), none per step: the
    state lives in VGPRs / AGPRs / LDS.  The same bound holds for the "pi" instantiations (per-instance tables through LDS-DMA slots: 3 / 34 with
    both tables, 3 / 34 bounds only, 5 / 33 reference only) — the register allocator sits at a cliff there: with the lanes' DMA addresses kept the
    other way round (remade per iteration instead of carried) the warm-start instantiations pick up 170 - 300 accesses per iteration."""
    seen = 0
    for src in ("admm_tile16.hip", "admm_tile16_pi.hip"):
        for name, lines in kernels_of(listings[src]).items():
            m = re.search(r"admm_tile16_kernelILi30ELb1ELb([01])ELb([01])ELb([01])ELb([01])E", name)   # exact arithmetic; COLD, MPC, BR, XR
            if not m:
                continue
            a, b = iteration_loop(lines)
            n_scratch = sum(1 for l in lines[a:b] if l.startswith("scratch_"))
            n_mfma = sum(1 for l in lines[a:b] if l.startswith("v_mfma"))
            mpc = m.group(2) == "1"   # the closed loop on chip keeps a few more values live across the loop
            assert n_mfma >= 29 * 9 and n_scratch <= (60 if mpc else 45), (name, n_mfma, n_scratch)
            seen += 1
    assert seen == 4 + 6


def test_tile16_pi_kernels_own_m0_and_count_their_dma(listings):
    """The LDS-DMA statements of the "pi" instantiations write M0 without saving it (a lone wave pays every scalar instruction with an issue
    slot): nothing else in those kernels may touch M0.  And the iteration loop must issue exactly the DMAs the design counts — per iteration
    N - 1 reference rows (XR) and 2 N bounds pieces (BR) — and no other vector memory instruction."""
    N = 30
    seen = 0
    for name, lines in kernels_of(listings["admm_tile16_pi.hip"]).items():
        m = re.search(r"admm_tile16_kernelILi30ELb[01]ELb[01]ELb0ELb([01])ELb([01])E", name)
        if not m:
            continue
        br, xr = m.group(1) == "1", m.group(2) == "1"
        for l in lines:
            if re.search(r"\bm0\b", l):
                assert re.match(r"s_mov_b32 m0, s\d+$|s_add_u32 m0, m0, 0x3f0$", l), (name, l)   # (the second: piece 2 of a resident {lo, hi} row, per tile)
        a, b = iteration_loop(lines)
        n_dma = sum(1 for l in lines[a:b] if l.startswith("global_load_lds_dwordx4"))
        assert n_dma == (2 * N if br else 0) + (N - 1 if xr else 0), (name, n_dma)
        other = [l for l in lines[a:b] if re.match(r"(global|buffer|flat)_", l) and not l.startswith("global_load_lds_dwordx4")]
        assert not other, (name, other[:3])
        seen += 1
    assert seen == 12
