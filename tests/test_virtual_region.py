"""the layout of halo2-base's virtual cells into physical columns (halo2_lib_amd/virtual_region.py restating
halo2-base/src/gates/flex_gate/threads/single_phase.rs:193-263 (keygen) and :273-312 (assign_witnesses), virtual_region/lookups.rs:129-156,
virtual_region/copy_constraints.rs:120-173, gates/circuit/mod.rs:159-203).

  * the layout rules on hand-made threads (break points, the duplicated break cell and its copy constraint, overlap safety, running out of
    columns), and that assign_witnesses reproduces the keygen-stage columns from the break points alone;
  * circuits built through a tiny GateChip / RangeChip on these rules — several gate columns, overlapping inner-product gates across a break,
    lookup-advice columns or the single-column q_lookup form, constants, instances, two threads — go through keygen and create_proof on the
    emulated build (CPU suite) and on the GPU: proof BYTES equal the oracle prover's on the same columns, both verifiers accept, and a witness
    that violates the copy constraint of a break cell is rejected."""
import random

import numpy as np
import pytest

import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP
from halo2_lib_amd import plonk as PL
from halo2_lib_amd import virtual_region as VR
from oracle import bn254 as O
from oracle import plonk as P
from tests.util import PreDrawnRng, R


# ---------------------------------------------------------------------------------------------------------------- layout rules (no prover)
def _thread(cm, ctx_id, cells, gates):
    ctx = VR.Context(False, VR.FIRST_PHASE_TYPE_ID, ctx_id, cm)
    ctx.assign_region([VR.Witness(v) for v in cells], gates)
    return ctx


def test_break_points_and_duplicated_break_cell():
    cm = VR.CopyConstraintManager()
    # 3 chained gates (offsets 0, 3, 6) then a free cell: 11 cells; max_rows = 8 -> the gate at row 6 does not fit (6 + 4 > 8)
    ctx = _thread(cm, 0, list(range(100, 111)), [0, 3, 6])
    region = VR.Region(16, 2)
    bps = VR.assign_with_constraints([ctx], [0, 1], region, cm, max_rows=8)
    assert bps == [6]
    assert [region.advice[0][r] for r in range(7)] == list(range(100, 107))           # rows 0..6 of column 0, the cell at the break included
    assert [region.advice[1][r] for r in range(5)] == list(range(106, 111))           # the break cell AGAIN at row 0 of the next column
    assert region.copies == [((("advice", 1), 0), (("advice", 0), 6))]                # (new cell, old cell): raw_constrain_equal(ncell, cell)
    assert region.selectors["q_enable"] == {0: {0, 3}, 1: {0}}                         # the third gate is enabled in the new column
    assert cm.assigned_advices[VR.ContextCell(VR.FIRST_PHASE_TYPE_ID, 0, 6)] == (("advice", 0), 6)   # the virtual cell keeps its first raw cell
    # proving side: same columns from the break points alone
    again = VR.Region(16, 2)
    VR.assign_witnesses([ctx], [0, 1], again, bps)
    assert again.advice == region.advice and again.copies == []


def test_break_at_last_usable_row_without_gate():
    cm = VR.CopyConstraintManager()
    ctx = _thread(cm, 0, list(range(1, 13)), [])       # 12 cells, no gates: a column is left at row max_rows - 1
    region = VR.Region(8, 3)
    bps = VR.assign_with_constraints([ctx], [0, 1, 2], region, cm, max_rows=5)
    assert bps == [4, 4]
    assert [sorted(c) for c in region.advice] == [[0, 1, 2, 3, 4], [0, 1, 2, 3, 4], [0, 1, 2, 3]]
    assert region.advice[1][0] == region.advice[0][4] == 5 and region.advice[2][0] == region.advice[1][4] == 9
    assert len(region.copies) == 2


def test_not_enough_columns_and_overlap_safety():
    cm = VR.CopyConstraintManager()
    ctx = _thread(cm, 0, list(range(20)), [0, 3, 6, 9])
    with pytest.raises(RuntimeError, match="NOT ENOUGH ADVICE COLUMNS"):
        VR.assign_with_constraints([ctx], [0], VR.Region(16, 1), cm, max_rows=8)
    cm = VR.CopyConstraintManager()
    bad = _thread(cm, 0, list(range(12)), [0, 4, 5])   # gates at 4 and 5 overlap with delta = 1: unsupported at a break
    with pytest.raises(AssertionError, match="overlaps"):
        VR.assign_with_constraints([bad], [0, 1], VR.Region(16, 2), cm, max_rows=8)
    # two threads share a column; an empty thread is skipped
    cm = VR.CopyConstraintManager()
    t0, t1, t2 = _thread(cm, 0, [1, 2, 3, 4], [0]), VR.Context(False, VR.FIRST_PHASE_TYPE_ID, 1, cm), _thread(cm, 2, [5, 6, 7, 8], [0])
    region = VR.Region(16, 1)
    assert VR.assign_with_constraints([t0, t1, t2], [0], region, cm, max_rows=12) == []
    assert [region.advice[0][r] for r in range(8)] == [1, 2, 3, 4, 5, 6, 7, 8] and region.selectors["q_enable"] == {0: {0, 4}}


def test_lookup_and_constant_assignment_order():
    cm = VR.CopyConstraintManager()
    lm = VR.LookupAnyManager(False, cm)
    a = _thread(cm, 1, [7, 8], [])
    b = _thread(cm, 0, [3], [])
    region = VR.Region(16, 4)
    VR.assign_with_constraints([b, a], [0], region, cm, max_rows=12)
    lm.add_lookup(a.tag(), a.get(0))
    lm.add_lookup(a.tag(), a.get(1))
    lm.add_lookup(b.tag(), b.get(0))
    lm.assign_raw([2, 3], region)          # tag order (context 0 first), left to right, then top to bottom
    assert (region.advice[2], region.advice[3]) == ({0: 3, 1: 8}, {0: 7})
    assert region.copies == [((("advice", 0), 0), (("advice", 2), 0)), ((("advice", 0), 1), (("advice", 3), 0)), ((("advice", 0), 2), (("advice", 2), 1))]
    # constants: sorted by value, assigned left to right then top to bottom, one fixed cell per distinct value
    cm2 = VR.CopyConstraintManager()
    c = VR.Context(False, VR.FIRST_PHASE_TYPE_ID, 0, cm2)
    for v in (9, 2, 9, 5):
        c.load_constant(v)
    region2 = VR.Region(16, 1)
    VR.assign_with_constraints([c], [0], region2, cm2, max_rows=12)
    VR.copy_manager_assign_raw(cm2, [4, 5], region2)
    assert region2.fixed == {4: {0: 2, 1: 9}, 5: {0: 5}}
    assert region2.copies == [((("fixed", 4), 0), (("advice", 0), 1)), ((("fixed", 5), 0), (("advice", 0), 3)),
                              ((("fixed", 4), 1), (("advice", 0), 0)), ((("fixed", 4), 1), (("advice", 0), 2))]


# ---------------------------------------------------------------------------------------------------------------- circuits through the prover
def _program(builder: VR.BaseCircuitBuilder, seed: int, gates: int, lookup_bits, num_instance: int):
    """the same deterministic gadget program for the keygen and the prover stage (the reference runs its circuit closure twice as well)"""
    g = random.Random(seed)
    rng_chip = VR.RangeChip(lookup_bits, builder.lookup_manager) if lookup_bits is not None else None
    ctx = builder.main()
    x = ctx.load_witness(g.randrange(R))
    acc = x
    outs = []
    for j in range(gates):
        if j == gates // 2:
            ctx = builder.new_thread()          # a second thread: its cells follow the first thread's in the same columns
            acc = ctx.load_witness(acc.value)
        kind = j % 5
        if kind == 0:
            acc = VR.GateChip.mul_add(ctx, acc, VR.Witness(g.randrange(R)), VR.Constant(g.randrange(1, 6)))
        elif kind == 1:
            acc = VR.GateChip.inner_product(ctx, [acc, VR.Witness(g.randrange(R)), VR.Witness(g.randrange(2)), VR.Witness(g.randrange(R))],
                                            [VR.Witness(g.randrange(R)), VR.Constant(5), VR.Witness(g.randrange(R)), VR.Witness(1)])
        elif kind == 2 and rng_chip is not None:
            small = ctx.load_witness(g.randrange(1 << lookup_bits))
            rng_chip.add_cell_to_lookup(ctx, small)
            acc = VR.GateChip.add(ctx, acc, small)
        elif kind == 3:
            acc = VR.GateChip.mul(ctx, acc, acc)
        else:
            acc = VR.GateChip.add(ctx, acc, VR.Constant(1))
        if j % 7 == 3:
            outs.append(acc)
    builder.assigned_instances = [[x, acc] + outs[:2]] + [[outs[-1]]] * (num_instance - 1) if num_instance else []


def _run(ctx, k, na, nl, nf, ni, lb, gates, seed=5, threads=2):
    sh = P.Shape(k, na, nl, nf, ni, lb)
    kb = VR.BaseCircuitBuilder(False)
    _program(kb, seed, gates, lb, ni)
    advice_k, fixed, copies, instances, bps = kb.synthesize(sh)
    assert len(bps) >= 1, "the test circuit must cross at least one break point"
    total_advice, total_fixed, total_lookup = kb.statistics()
    assert kb.calculate_params(k, sh.blinding_factors + 3, lb, ni)[1] <= na
    # prover stage: witness generation only, columns from the break points
    pb = VR.BaseCircuitBuilder(True)
    pb.break_points = bps
    _program(pb, seed, gates, lb, ni)
    advice, _, _, inst2, _ = pb.synthesize(sh)
    assert all(np.array_equal(a, b) for a, b in zip(advice, advice_k)) and all(np.array_equal(a, b) for a, b in zip(instances, inst2))
    # every gate row satisfies a + b*c = d, every copy holds (sanity of the layout itself, independent of any prover)
    vals = [O.limbs_to_ints(c, R) for c in advice]
    fvals = [O.limbs_to_ints(c, R) for c in fixed]
    for gi, col in enumerate(sh.gate_advice):
        q = fvals[sh.q_enable_cols[gi]]
        for r in range(sh.n - (sh.blinding_factors + 3) - 3):
            if q[r]:
                assert (vals[col][r] + vals[col][r + 1] * vals[col][r + 2] - vals[col][r + 3]) % R == 0, (gi, r)
    cell = lambda c: {"advice": vals, "fixed": fvals, "instance": [O.limbs_to_ints(v, R) for v in instances]}[c[0][0]][c[0][1]][c[1]]
    assert all(cell(l) == cell(r) for l, r in copies)
    # keygen + proof on the backend, against the oracle prover on the same columns
    s_toxic = 0xA5A5 + seed
    kzg = HP.ParamsKZG.setup(ctx, k, s_toxic, precompute=False)
    params = P.Params.setup(k, s_toxic, g=ctx.bases_download(kzg.g), g_lagrange=ctx.bases_download(kzg.g_lagrange))
    gpk = PL.keygen(kzg, PL.BaseCircuitParams.new(k, na, nl, nf, ni, lb), fixed, copies)
    asm = P.PermutationAssembly(sh)
    for l, r in copies:
        asm.copy(l, r)
    opk = P.keygen(params, sh, fixed, asm, threads)
    assert gpk.transcript_repr == opk.vk.transcript_repr
    budget = 4096 + (1 << k)
    proof = PL.create_proof(gpk, advice, instances, PreDrawnRng(budget, 11))
    inst_ints = [O.limbs_to_ints(v, R) for v in instances]
    assert proof == P.create_proof(params, opk, advice, inst_ints, PreDrawnRng(budget, 11), threads), "proof bytes differ from the oracle prover's"
    assert P.verify_proof(params, opk.vk, inst_ints, proof) and PL.verify_proof(gpk, instances, proof)
    # a witness whose duplicated break cell differs from the original violates the break's copy constraint: rejected
    (_, dup_col), dup_row = copies[0][0][0], copies[0][0][1]
    assert copies[0][0][0][0] == "advice" and dup_row == 0
    bad = [np.array(c) for c in advice]
    bad[dup_col][0] = O.ints_to_limbs([(vals[dup_col][0] + 1) % R], R)[0]
    forged = PL.create_proof(gpk, bad, instances, PreDrawnRng(budget, 12))
    assert not PL.verify_proof(gpk, instances, forged)
    gpk.free()
    kzg.free()
    return sh, bps


@pytest.mark.parametrize("shape,gates", [((6, 3, 1, 1, 1, 4), 26), ((6, 1, 1, 1, 0, 4), 7)])
def test_layout_proves_emulated(shape, gates):
    """(k=6: 55 rows for the layout) three gate columns + a lookup-advice column + instances; and the single-column q_lookup form"""
    from tests.emu_util import emu_context

    ctx = emu_context()
    try:
        if shape[1] == 1:
            with pytest.raises(RuntimeError, match="NOT ENOUGH ADVICE COLUMNS"):
                _run(ctx, *shape, gates=40)
            sh = P.Shape(*shape)
            kb = VR.BaseCircuitBuilder(False)
            _program(kb, 5, gates, shape[5], 0)
            advice, fixed, copies, instances, bps = kb.synthesize(sh)
            assert bps == [] and any(O.limbs_to_ints(fixed[sh.q_lookup_col], R))    # the lookup sits behind q_lookup on the gate column
            kzg = HP.ParamsKZG.setup(ctx, shape[0], 77, precompute=False)
            gpk = PL.keygen(kzg, PL.BaseCircuitParams.new(*shape), fixed, copies)
            proof = PL.create_proof(gpk, advice, instances, PreDrawnRng(4096, 3))
            assert PL.verify_proof(gpk, instances, proof)
            gpk.free()
            kzg.free()
        else:
            sh, bps = _run(ctx, *shape, gates=gates)
            assert len(bps) == 2
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,gates", [((10, 4, 1, 1, 2, 8), 520), ((9, 2, 2, 2, 1, 7), 130)])
def test_layout_proves_gpu(shape, gates):
    ctx = H.Context()
    try:
        sh, bps = _run(ctx, *shape, gates=gates, threads=8)
        assert len(bps) == shape[1] - 1
    finally:
        ctx.close()
