"""GateInstructions' witness values on a backend of F_r batch operations — `add`, `sub`, `mul`, `mul_add`, `invert` over
(n, 4) Montgomery limb arrays: libh2hip's h2hip_fr_*_batch_dev kernels on the GPU, the C oracle on the CPU.  Each operation is computed the way
the reference computes the assigned value (halo2-base/src/gates/flex_gate/mod.rs): add :158-175, sub :184-201, sub_mul :213-231, neg :233-244,
mul :246-263, mul_add :265-283, mul_not :285-301, div_unsafe :315-338 (a * b^-1), inner_product :346 / :994, inner_product_with_sums :400 / :1091,
sum_products_with_coeff_and_var :512 / :1115, and :555, not :569, select :580 / :1144 (sel * (a - b) + b), or_and :595 / :1179 (a + b c - a b c),
pow_var :840 (square-and-multiply over the exponent's bits)."""
import json
import os

from tests.util import R, fr

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    return json.load(open(os.path.join(HERE, "golden", "flex_gate_reference_kats.json")))["cases"]


def expected_int(e):
    if isinstance(e, dict):
        return (R - e["neg"]) % R
    return e


def evaluate(op, inputs, B):
    """-> list of result limb rows (most operations: one)"""
    f = lambda v: fr([v % R])
    one, zero = f(1), f(0)
    if op == "add":
        return [B.add(f(inputs[0]), f(inputs[1]))]
    if op == "inc":
        return [B.add(f(inputs[0]), one)]
    if op == "sub":
        return [B.sub(f(inputs[0]), f(inputs[1]))]
    if op == "dec":
        return [B.sub(f(inputs[0]), one)]
    if op == "sub_mul":
        return [B.sub(f(inputs[0]), B.mul(f(inputs[1]), f(inputs[2])))]
    if op == "neg":
        return [B.sub(zero, f(inputs[0]))]
    if op == "mul" or op == "and":
        return [B.mul(f(inputs[0]), f(inputs[1]))]
    if op == "mul_add":
        return [B.mul_add(f(inputs[0]), f(inputs[1]), f(inputs[2]))]
    if op == "mul_not":
        return [B.mul(B.sub(one, f(inputs[0])), f(inputs[1]))]
    if op == "not":
        return [B.sub(one, f(inputs[0]))]
    if op == "div_unsafe":
        return [B.mul(f(inputs[0]), B.invert(f(inputs[1])))]
    if op in ("inner_product", "inner_product_left", "inner_product_with_sums"):
        acc, sums = zero, []
        for a, b in zip(*inputs):
            acc = B.mul_add(f(a), f(b), acc)
            sums.append(acc)
        return sums if op == "inner_product_with_sums" else [acc]
    if op == "sum_products_with_coeff_and_var":
        acc = f(inputs[1])
        for c, a, b in inputs[0]:
            acc = B.mul_add(B.mul(f(c), f(a)), f(b), acc)
        return [acc]
    if op == "select":   # select(a, b, sel)
        a, b, sel = (f(v) for v in inputs)
        return [B.mul_add(sel, B.sub(a, b), b)]
    if op == "or_and":
        a, b, c = (f(v) for v in inputs)
        bc = B.mul(b, c)
        return [B.sub(B.add(a, bc), B.mul(a, bc))]
    if op == "pow_var":
        a, e = f(inputs[0]), inputs[1]
        acc = one
        for bit in bin(e)[2:]:
            acc = B.mul(acc, acc)
            if bit == "1":
                acc = B.mul(acc, a)
        return [acc]
    raise KeyError(op)


def check_all(B):
    import numpy as np

    n = 0
    for c in load():
        got = evaluate(c["op"], c["inputs"], B)
        exp = c["expected"] if isinstance(c["expected"], list) else [c["expected"]]
        assert len(got) == len(exp), c
        for g, e in zip(got, exp):
            assert np.array_equal(np.asarray(g).reshape(1, 4), fr([expected_int(e)])), (c, g)
        n += 1
    return n
