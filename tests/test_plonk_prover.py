"""create_proof for halo2-base circuits (include/h2hip.h "a1", halo2-lib_amd/csrc/plonk.hip) against the oracle's restatement
(oracle/plonk.py) of what the reference runs at halo2-base/src/utils/testing.rs:32-50 (prove) and :64-88 (verify):

  * proof BYTES equal to the oracle prover's on the same SRS, circuit and RNG stream (BASELINE.json: "proof bytes equal to CPU");
  * the oracle's verifier — transcript replay, quotient identity at x, SHPLONK with a real pairing — accepts the HIP proof and
    rejects tampered ones;
  * shapes: the k=19 ECDSA configuration of halo2-ecc/configs/secp256k1/bench_ecdsa.config:1 (1 advice, lookup on the gate column
    behind q_lookup, 1 constants column, no instances), multi-column shapes with dedicated lookup-advice columns (the form of
    halo2-ecc/configs/bn254/bench_pairing.config), instance columns, and circuits without a range table.
"""
import numpy as np
import pytest

import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP
from halo2_lib_amd import plonk as PL
from halo2_lib_amd import testing as T
from oracle import bn254 as O
from oracle import c_oracle as CO
from oracle import plonk as P
from tests.util import PreDrawnRng, R


class _OracleBackend:
    mul = staticmethod(CO.fr_mul)
    add = staticmethod(CO.fr_add)


def _rng_budget(sh):
    """how many Fr::random draws one proof consumes"""
    n, bf = sh.n, sh.blinding_factors
    return (sh.num_advice_total * (bf + 2) + len(sh.lookups) * (2 * (bf + 1) + 2 + bf + 1) + sh.num_perm_sets * (bf + 1) + n + 1 +
            sh.quotient_poly_degree + 16)


def _setup(ctx, k, na, nl, nf, ni, lb, seed, threads, precompute):
    sh = P.Shape(k, na, nl, nf, ni, lb)
    s_toxic = 0x1D0C0FFEE1234567890ABCDEF + seed
    kzg = HP.ParamsKZG.setup(ctx, k, s_toxic, precompute=precompute)
    # the oracle uses the SAME SRS (downloaded; the GPU setup itself is tested against the definition in test_host_mirror.py)
    params = P.Params.setup(k, s_toxic, g=ctx.bases_download(kzg.g), g_lagrange=ctx.bases_download(kzg.g_lagrange))
    circ = T.build_circuit(sh, seed, _OracleBackend)
    bp = PL.BaseCircuitParams.new(k, na, nl, nf, ni, lb)
    gpk = PL.keygen(kzg, bp, circ.fixed, circ.copies)
    return sh, kzg, params, circ, gpk


def _oracle_pk(sh, params, circ, threads):
    asm = P.PermutationAssembly(sh)
    for l, r in circ.copies:
        asm.copy(l, r)
    return P.keygen(params, sh, circ.fixed, asm, threads)


def _vk_from_gpu(sh, gpk):
    """a VerifyingKey for the oracle verifier from the HIP keygen's commitments (large k: the oracle keygen is skipped)"""
    pts = lambda a: O.limbs_to_points(np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 8))
    return P.VerifyingKey(sh, pts(gpk.fixed_commitments), pts(gpk.permutation_commitments) if len(gpk.permutation_commitments) else [],
                          gpk.transcript_repr)


def _check(ctx, k, na, nl, nf, ni, lb, seed=3, threads=4, oracle_prover=True, precompute=False, second_proof=True):
    sh, kzg, params, circ, gpk = _setup(ctx, k, na, nl, nf, ni, lb, seed, threads, precompute)
    try:
        shape = gpk.shape
        assert (shape.degree, shape.extended_k, shape.blinding_factors, shape.usable_rows, shape.num_perm_sets, shape.num_fixed_total) == (
            sh.degree, sh.extended_k, sh.blinding_factors, sh.usable_rows, sh.num_perm_sets, sh.num_fixed_total)
        inst = [O.limbs_to_ints(v, R) for v in circ.instances]
        budget = _rng_budget(sh)
        timings = {}
        got = PL.create_proof(gpk, circ.advice, circ.instances, PreDrawnRng(budget, 1000 + seed), timings)
        assert len(got) == gpk.proof_size() and set(timings) == {ctx.lib.h2hip_plonk_stage_name(i).decode() for i in range(PL.PLONK_STAGES)}
        if oracle_prover:
            pk = _oracle_pk(sh, params, circ, threads)
            assert gpk.transcript_repr == pk.vk.transcript_repr, "verifying keys differ (fixed / permutation commitments)"
            want = P.create_proof(params, pk, circ.advice, inst, PreDrawnRng(budget, 1000 + seed), threads)
            assert got == want, "proof bytes differ from the oracle prover's"
            vk = pk.vk
        else:
            vk = _vk_from_gpu(sh, gpk)
        assert P.verify_proof(params, vk, inst, got), "the oracle verifier rejects the HIP proof"
        assert PL.verify_proof(gpk, circ.instances, got), "libh2hip's own verifier rejects the proof"
        # libh2hip's own array RNG (h2hip_array_rng_fill, no Python frame per draw) serves the same stream: same bytes, position advanced alike
        ref_rng, arr_rng = PreDrawnRng(budget, 1000 + seed), None
        arr_rng = PL.ArrayRng(ref_rng.values)
        assert PL.create_proof(gpk, circ.advice, circ.instances, arr_rng) == got
        PL.create_proof(gpk, circ.advice, circ.instances, ref_rng)
        assert arr_rng.pos == ref_rng.pos > 0
        # a second proof from the same key (pooled buffers reused) with another RNG stream: different bytes, still valid
        if second_proof:
            again = PL.create_proof(gpk, circ.advice, circ.instances, PreDrawnRng(budget, 2000 + seed))
            assert again != got and P.verify_proof(params, vk, inst, again)
        return sh, params, vk, inst, got, circ, gpk, kzg
    except Exception:
        gpk.free()
        kzg.free()
        raise


def _tamper_checks(sh, params, vk, inst, proof, circ, gpk, forge=True):
    # an evaluation flipped -> the quotient identity or the opening fails
    first_eval = 32 * (sh.num_advice_total + 3 * len(sh.lookups) + sh.num_perm_sets + 1 + sh.quotient_poly_degree)
    bad = bytearray(proof)
    bad[first_eval] ^= 1
    try:
        assert not P.verify_proof(params, vk, inst, bytes(bad))
    except P.VerifyError:
        pass
    # a witness that violates the gate: the prover still produces bytes (like upstream), the verifier rejects them
    if forge:
        adv = [np.array(c) for c in circ.advice]
        adv[0][3] = CO.fr_add(adv[0][3:4], O.ints_to_limbs([1], R))[0]
        forged = PL.create_proof(gpk, adv, circ.instances, PreDrawnRng(_rng_budget(sh), 7))
        assert not P.verify_proof(params, vk, inst, forged)
    if inst and len(inst[0]):
        wrong = [list(v) for v in inst]
        wrong[0][0] = (wrong[0][0] + 1) % R
        assert not P.verify_proof(params, vk, wrong, proof)
        assert not PL.verify_proof(gpk, [O.ints_to_limbs(v, R) for v in wrong], proof)
    # libh2hip's verifier (h2hip_plonk_verify_proof: the reference's check_proof) agrees with the oracle's on mutated proofs
    g = np.random.default_rng(len(proof))
    for pos in [0, 31, first_eval + 5, len(proof) - 1] + [int(v) for v in g.integers(0, len(proof), size=6)]:
        mutated = bytearray(proof)
        mutated[pos] ^= 1 << int(g.integers(0, 8))
        try:
            want = P.verify_proof(params, vk, inst, bytes(mutated))
        except P.VerifyError:
            want = False
        assert PL.verify_proof(gpk, circ.instances, bytes(mutated)) == want == False, pos
    assert not PL.verify_proof(gpk, circ.instances, proof[:-32]) and not PL.verify_proof(gpk, circ.instances, proof + bytes(32))


SMALL_SHAPES = [(6, 1, 1, 1, 0, 4), (7, 2, 1, 1, 1, 5), (6, 1, 0, 1, 0, None), (6, 2, 2, 2, 1, 3),
                (5, 6, 4, 1, 1, 3)]   # the last: enough columns / lookups / permutation sets for every batched path (>= 4 of each)


@pytest.mark.parametrize("shape", SMALL_SHAPES)
def test_create_proof_emulated(shape):
    """the prover's host logic and kernels on the CPU-emulated build: byte equality, verification, tampering"""
    from tests.emu_util import emu_context

    ctx = emu_context()
    try:
        full = shape == SMALL_SHAPES[0]   # the emulated kernels are slow: the extra proofs run for one shape here, for all on the GPU
        sh, params, vk, inst, proof, circ, gpk, kzg = _check(ctx, *shape, threads=2, second_proof=full)
        _tamper_checks(sh, params, vk, inst, proof, circ, gpk, forge=full)
        gpk.free()
        kzg.free()
    finally:
        ctx.close()


def test_create_proof_argument_errors_emulated():
    from tests.emu_util import emu_context

    ctx = emu_context()
    try:
        sh, kzg, params, circ, gpk = _setup(ctx, 6, 1, 1, 1, 0, 4, 1, 1, False)
        with pytest.raises(ValueError):
            PL.create_proof(gpk, [], [], PreDrawnRng(8, 1))
        with pytest.raises(RuntimeError):      # the RNG runs dry: reported, nothing hangs
            PL.create_proof(gpk, circ.advice, [], PreDrawnRng(8, 1))
        with pytest.raises(RuntimeError):      # the same through the library's array RNG
            PL.create_proof(gpk, circ.advice, [], PL.ArrayRng(PreDrawnRng(8, 1).values))
        bad_copy = np.array([[0, sh.usable_rows, 1, 0]], dtype=np.uint32)   # a copy constraint in the blinding rows
        with pytest.raises(H.H2HipError):
            PL.keygen(kzg, PL.BaseCircuitParams.new(6, 1, 1, 1, 0, 4), circ.fixed, bad_copy)
        with pytest.raises(H.H2HipError):      # lookup table larger than the usable rows
            PL.shape_of(ctx, PL.BaseCircuitParams.new(6, 1, 1, 1, 0, 6))
        missing = [np.array(c) for c in circ.advice]
        missing[0][0] = O.ints_to_limbs([(1 << 4) + 5], R)[0]              # a range-checked cell outside the table
        with pytest.raises(H.H2HipError):
            PL.create_proof(gpk, missing, [], PreDrawnRng(_rng_budget(sh), 1))
        gpk.free()
        kzg.free()
    finally:
        ctx.close()


def test_create_proof_with_kernel_profile_emulated():
    """bench.py times proofs with the library's HIP-event profile enabled: brackets nest (the lookup permutation's contains the scan's), an
    open bracket must never poison the next launch, and the account names the prover's kernels"""
    from tests.emu_util import emu_context

    ctx = emu_context()
    try:
        sh, kzg, params, circ, gpk = _setup(ctx, 6, 2, 1, 1, 0, 4, 1, 1, False)
        plain = PL.create_proof(gpk, circ.advice, circ.instances, PreDrawnRng(_rng_budget(sh), 1))
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(2):
            assert PL.create_proof(gpk, circ.advice, circ.instances, PreDrawnRng(_rng_budget(sh), 1)) == plain
        ctx.profile_enable(False)
        acc = ctx.profile_dump()
        assert {"msm_accum_kernel", "ntt_pass_kernel", "lookup_permute_kernels", "scan_kernels"} <= set(acc), sorted(acc)
        assert all(cnt > 0 and ms >= 0 and busy >= 0 for ms, cnt, busy in acc.values())
        assert acc["msm_accum_kernel"][1] % 2 == 0
        gpk.free()
        kzg.free()
    finally:
        ctx.close()


def test_create_proof_schedule_switches_emulated():
    """the schedule switches of r05's last measurements give the same bytes: the grand products' lagrange_to_coeff queued in front of the round's
    commitments (plonk_early_intt), the lanes' first sorts one behind the other (msm_stagger_sorts; msm_fuse_cols = 1 takes the per-column lanes the
    large sizes use), the quotient's gate identities in front of the join of the grand products' transforms (plonk_gate_before_join); precomputed
    window tables so that the batch path with its deferred reduction runs"""
    from tests.emu_util import emu_context

    ctx = emu_context()
    try:
        sh, kzg, params, circ, gpk = _setup(ctx, 6, 2, 2, 1, 0, 4, 1, 1, True)
        rng = lambda: PreDrawnRng(_rng_budget(sh), 1)
        plain = PL.create_proof(gpk, circ.advice, circ.instances, rng())
        names = ("plonk_early_intt", "msm_stagger_sorts", "msm_fuse_cols", "plonk_gate_before_join")
        old = {n: ctx.get_param(n) for n in names}
        try:
            for vals in ((1, 0, 0, 0), (0, 1, 1, 1), (1, 1, 1, 1), (0, 0, 0, 0)):
                for n, v in zip(names, vals):
                    ctx.set_param(n, v)
                assert PL.create_proof(gpk, circ.advice, circ.instances, rng()) == plain, vals
        finally:
            for n, v in old.items():
                ctx.set_param(n, v)
        gpk.free()
        kzg.free()
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(9, 1, 1, 1, 0, 8), (12, 1, 1, 1, 1, 11), (12, 2, 1, 1, 1, 11), (13, 4, 2, 2, 2, 10), (10, 1, 0, 1, 0, None),
                                   (11, 20, 4, 2, 1, 10), (9, 1, 1, 0, 0, 7), (9, 2, 0, 0, 1, None), (10, 1, 2, 1, 0, 8)])   # (11, 20, 4, ...): a wide shape like the reference's low-k configurations (13 chained permutation sets); then: no constants column, no range chip with an instance column, one advice column with num_lookup_advice > 1
def test_create_proof_gpu(shape):
    ctx = H.Context()
    try:
        sh, params, vk, inst, proof, circ, gpk, kzg = _check(ctx, *shape, threads=8, precompute=shape[0] >= 12)
        _tamper_checks(sh, params, vk, inst, proof, circ, gpk)
        gpk.free()
        kzg.free()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_create_proof_gpu_very_wide_shape():
    """a shape past every batching boundary at once — 70 gate columns (> 64 per quotient launch, > 32 per NTT launch), 34 lookups (> 32 per
    launch), 36 permutation sets (> 12 (set, term) jobs per launch), > 64 commitments in one round (deferred reduction in chunks) — like the
    reference's low-k configurations (bench_ecdsa.config: k = 11, 291 + 53 columns).  Byte equality with the oracle prover."""
    ctx = H.Context()
    try:
        sh, params, vk, inst, proof, circ, gpk, kzg = _check(ctx, 8, 70, 34, 2, 1, 5, threads=16, precompute=True, second_proof=False)
        assert sh.num_perm_sets > 24 and len(sh.lookups) == 34
        gpk.free()
        kzg.free()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_create_proof_gpu_k16_bytes_equal():
    ctx = H.Context()
    try:
        *_, gpk, kzg = _check(ctx, 16, 1, 1, 1, 0, 15, threads=16, precompute=True)
        gpk.free()
        kzg.free()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_create_proof_gpu_k19_ecdsa_shape():
    """BASELINE.json configs[3]: the k = 19 secp256k1-ECDSA configuration (halo2-ecc/configs/secp256k1/bench_ecdsa.config:1).  Proof BYTES
    equal to the oracle prover's (its C kernels on the host cores: a few seconds), then the oracle's and libh2hip's verifiers."""
    ctx = H.Context()
    try:
        sh, params, vk, inst, proof, circ, gpk, kzg = _check(ctx, 19, 1, 1, 1, 0, 18, threads=32, oracle_prover=True, precompute=True)
        assert (sh.degree, sh.extended_k, gpk.shape.num_commitments) == (5, 21, 12)
        gpk.free()
        kzg.free()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_create_proof_repeatable_with_interleaved_keys():
    """race / state-leak detector (tools/soak.py in small): two proving keys of different shapes share one context; their proofs,
    alternated, must stay byte-identical to the first ones for a fixed RNG stream (pooled buffers, copy stream, scratch reuse)"""
    ctx = H.Context()
    try:
        a = _setup(ctx, 14, 1, 1, 1, 0, 13, 5, 8, True)
        b = _setup(ctx, 12, 3, 1, 1, 1, 11, 6, 8, True)
        (sha, _, _, ca, pka), (shb, _, _, cb, pkb) = a, b
        ra = PL.create_proof(pka, ca.advice, ca.instances, PreDrawnRng(_rng_budget(sha), 1))
        rb = PL.create_proof(pkb, cb.advice, cb.instances, PreDrawnRng(_rng_budget(shb), 2))
        assert PL.verify_proof(pka, ca.instances, ra) and PL.verify_proof(pkb, cb.instances, rb)
        for _ in range(12):
            assert PL.create_proof(pka, ca.advice, ca.instances, PreDrawnRng(_rng_budget(sha), 1)) == ra
            assert PL.create_proof(pkb, cb.advice, cb.instances, PreDrawnRng(_rng_budget(shb), 2)) == rb
        for pk, kzg in ((pka, a[1]), (pkb, b[1])):
            pk.free()
            kzg.free()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_create_proof_gpu_three_contexts_in_flight():
    """the C ABI's threading contract (include/h2hip.h: one context per thread, no shared mutable state): three host threads, each with its own
    context, window tables and proving key of the same k = 16 circuit, prove concurrently on the one GPU — every proof has the bytes of the proof
    made alone, which the oracle verifier accepts (tools/two_in_flight.py measures the throughput of this use)"""
    import threading

    provers = []
    try:
        for _ in range(3):
            ctx = H.Context()
            provers.append((ctx,) + _setup(ctx, 16, 2, 1, 1, 0, 15, 4, 8, True))
        sh = provers[0][1]
        rng = lambda: PreDrawnRng(_rng_budget(sh), 9)
        alone = PL.create_proof(provers[0][5], provers[0][4].advice, provers[0][4].instances, rng())
        got, errs = [[] for _ in provers], []
        gate = threading.Barrier(len(provers))

        def work(i):
            try:
                _, _, _, _, circ, pk = provers[i]
                gate.wait()
                for _ in range(6):
                    got[i].append(PL.create_proof(pk, circ.advice, circ.instances, rng()))
            except BaseException as e:   # noqa: BLE001 — reported by the parent
                errs.append((i, repr(e)))

        ths = [threading.Thread(target=work, args=(i,)) for i in range(len(provers))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errs, errs
        assert all(len(g) == 6 and all(p == alone for p in g) for g in got)
        assert P.verify_proof(provers[0][3], _vk_from_gpu(sh, provers[0][5]), [], alone)
    finally:
        for pr in provers:
            pr[5].free()
            pr[2].free()
            pr[0].close()


@pytest.mark.gpu
def test_create_proof_gpu_k21_pairing_shape():
    """BASELINE.json configs[4] on one GPU: the k = 21 BN254-pairing configuration (halo2-ecc/configs/bn254/bench_pairing.config:8: 2 gate advice
    columns, 1 lookup-advice column, 1 constants column, lookup_bits 20 -> degree 4, extended_k 23, 14 commitments of 2^21 points): proof BYTES
    equal to the oracle prover's (about half a minute of C kernels on the host cores), then the oracle's and libh2hip's verifiers"""
    ctx = H.Context()
    try:
        sh, params, vk, inst, proof, circ, gpk, kzg = _check(ctx, 21, 2, 1, 1, 0, 20, threads=32, oracle_prover=True, precompute=True, second_proof=False)
        assert (sh.degree, sh.extended_k, gpk.shape.num_commitments) == (4, 23, 14)
        gpk.free()
        kzg.free()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_create_proof_gpu_random_shapes():
    """tools/fuzz_shapes.py for 25 s with a fixed seed: randomly drawn shapes (narrow / wide, with and without lookups, instances, precomputed
    bases), proof bytes equal to the oracle prover's for every one (a longer run: profiles/archive/r03_fuzz_shapes.log)"""
    import subprocess, sys, os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_shapes.py"), "25", "11"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "proof bytes equal to the oracle prover's for every one" in r.stdout
