"""
Host-side mirror of halo2-base's virtual-region layout: how a `BaseCircuitBuilder`'s virtual `Context` cell streams become the physical
columns, selector rows and copy constraints that `keygen_*` / `create_proof` (plonk.py -> libh2hip) take.  This is the code behind
INTEGRATION.md's statement that circuits written against halo2-base lay out unchanged: the prover's input is exactly what the reference's
`synthesize` writes into its region.

Restated from (file:line in /root/reference/halo2-base/src):

  lib.rs:91-104, 225-262            ContextCell (ordered by (type_id, context_id, offset)), Context::assign_cell
  lib.rs:306-334, 430-452           Context::assign_region, load_witness / load_constant
  gates/flex_gate/threads/single_phase.rs:193-263   assign_with_constraints  (keygen: break points, selectors, the break cell's copy)
  gates/flex_gate/threads/single_phase.rs:273-312   assign_witnesses         (proving: the same layout from the recorded break points)
  virtual_region/lookups.rs:129-156 LookupAnyManager::assign_raw   (cells to look up: left to right, then top to bottom)
  virtual_region/copy_constraints.rs:120-173        CopyConstraintManager::assign_raw (constants sorted, then advice / constant equalities)
  gates/circuit/builder.rs:260-288, 291-309, 327-375  calculate_params, assign_instances, assign_lookups_in_phase
  gates/circuit/mod.rs:159-203      BaseCircuitBuilder::synthesize (order: gate threads, lookups, copy manager, instances)
  gates/range/mod.rs:154-170        load_lookup_table
  gates/flex_gate/mod.rs:158-168, 246-277, 346-353, 1149-1190 (inner_product_simple)   the gate layouts used by the tests' tiny GateChip

First phase only (h2hip_base_circuit_params is single-phase: include/h2hip.h).  Values are canonical integers mod r here; `synthesize`
returns Montgomery-limb columns.  [UPSTREAM-RECALL]: `F: Ord` compares canonical representations numerically; `constrain_instance(cell, col, row)`
records the copy as (advice cell, instance cell).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FIRST_PHASE_TYPE_ID = "halo2-base:SinglePhaseCoreManager:FirstPhase"     # single_phase.rs:114-121
EXTERNAL_CELL_TYPE_ID = "halo2-base:External Raw Halo2 Cell"              # copy_constraints.rs:19
ROTATIONS = 4                                                             # the vertical gate q * (a + b * c - d) spans four rows


@dataclass(frozen=True, order=True)
class ContextCell:
    type_id: str
    context_id: int
    offset: int


@dataclass
class AssignedValue:
    value: int
    cell: Optional[ContextCell] = None


# QuantumCell variants (lib.rs:43-89)
@dataclass
class Existing:
    cell: AssignedValue

    @property
    def value(self):
        return self.cell.value


@dataclass
class Witness:
    value: int


@dataclass
class Constant:
    value: int


RawCell = Tuple[Tuple[str, int], int]      # ((kind, column index), row) — the copy-constraint endpoints plonk.keygen takes


@dataclass
class CopyConstraintManager:
    advice_equalities: List[Tuple[ContextCell, ContextCell]] = field(default_factory=list)
    constant_equalities: List[Tuple[int, ContextCell]] = field(default_factory=list)
    assigned_advices: Dict[ContextCell, RawCell] = field(default_factory=dict)
    assigned_constants: Dict[int, RawCell] = field(default_factory=dict)


class Context:
    """lib.rs:157-262: one virtual column of advice cells with its selector bits"""

    def __init__(self, witness_gen_only: bool, type_id: str, context_id: int, copy_manager: CopyConstraintManager, phase: int = 0):
        self.witness_gen_only, self.type_id, self.context_id, self.copy_manager, self.phase = witness_gen_only, type_id, context_id, copy_manager, phase
        self.advice: List[int] = []
        self.selector: List[bool] = []
        self.zero_cell: Optional[AssignedValue] = None

    def tag(self):
        return (self.type_id, self.context_id)

    def _latest_cell(self) -> ContextCell:
        return ContextCell(self.type_id, self.context_id, len(self.advice) - 1)

    def assign_cell(self, inp):
        if isinstance(inp, AssignedValue):
            inp = Existing(inp)
        if isinstance(inp, Existing):
            self.advice.append(inp.cell.value % R_MOD)
            if not self.witness_gen_only:
                self.copy_manager.advice_equalities.append((self._latest_cell(), inp.cell.cell))
        elif isinstance(inp, Witness):
            self.advice.append(inp.value % R_MOD)
        elif isinstance(inp, Constant):
            self.advice.append(inp.value % R_MOD)
            if not self.witness_gen_only:
                self.copy_manager.constant_equalities.append((inp.value % R_MOD, self._latest_cell()))
        else:
            raise TypeError(inp)

    def last(self) -> AssignedValue:
        return AssignedValue(self.advice[-1], None if self.witness_gen_only else self._latest_cell())

    def get(self, offset: int) -> AssignedValue:
        if offset < 0:
            offset += len(self.advice)
        assert 0 <= offset < len(self.advice)
        return AssignedValue(self.advice[offset], None if self.witness_gen_only else ContextCell(self.type_id, self.context_id, offset))

    def constrain_equal(self, a: AssignedValue, b: AssignedValue):
        if not self.witness_gen_only:
            self.copy_manager.advice_equalities.append((a.cell, b.cell))

    def assign_region(self, inputs, gate_offsets=()):
        row_offset = len(self.advice)
        for inp in inputs:
            self.assign_cell(inp)
        if not self.witness_gen_only:
            self.selector.extend([False] * (len(self.advice) - len(self.selector)))
            for off in gate_offsets:
                idx = row_offset + off
                assert 0 <= idx < len(self.selector), "Invalid selector offset"
                self.selector[idx] = True

    def assign_region_last(self, inputs, gate_offsets=()) -> AssignedValue:
        self.assign_region(inputs, gate_offsets)
        return self.last()

    def load_witness(self, v: int) -> AssignedValue:
        self.assign_region([Witness(v)])
        return self.last()

    def load_constant(self, c: int) -> AssignedValue:
        self.assign_region([Constant(c)])
        return self.last()

    def load_zero(self) -> AssignedValue:
        if self.zero_cell is None:
            self.zero_cell = self.load_constant(0)
        return self.zero_cell


# ---------------------------------------------------------------------------------------------------------------- physical layout
class Region:
    """the physical columns a `synthesize` writes: advice values, enabled selector rows, copy constraints in call order"""

    def __init__(self, n: int, num_advice: int):
        self.n = n
        self.advice: List[Dict[int, int]] = [dict() for _ in range(num_advice)]
        self.selectors: Dict[str, Dict[int, set]] = {}
        self.fixed: Dict[int, Dict[int, int]] = {}
        self.copies: List[Tuple[RawCell, RawCell]] = []

    def assign_advice(self, column: int, row: int, value: int) -> RawCell:
        assert 0 <= row < self.n, "row outside the circuit"
        self.advice[column][row] = value % R_MOD
        return (("advice", column), row)

    def assign_fixed(self, column: int, row: int, value: int) -> RawCell:
        assert 0 <= row < self.n
        self.fixed.setdefault(column, {})[row] = value % R_MOD
        return (("fixed", column), row)

    def enable(self, kind: str, column: int, row: int):
        self.selectors.setdefault(kind, {}).setdefault(column, set()).add(row)

    def constrain_equal(self, a: RawCell, b: RawCell):
        self.copies.append((a, b))


def assign_with_constraints(threads: Sequence[Context], gate_columns: Sequence[int], region: Region, copy_manager: CopyConstraintManager,
                            max_rows: int, rotations: int = ROTATIONS) -> List[int]:
    """single_phase.rs:193-263.  Keygen-side layout: walks the threads' cells down gate column 0, 1, ...; a column is left at the first row
    where the next gate would not fit (or at max_rows - 1), the cell at the break is assigned AGAIN at row 0 of the next column and the two
    are constrained equal (two gates may overlap in it).  Returns the break points (the row offsets at which columns were left)."""
    break_points: List[int] = []
    gate_index, row_offset = 0, 0
    for ctx in threads:
        if not ctx.advice:
            continue
        if gate_index >= len(gate_columns):
            raise RuntimeError("NOT ENOUGH ADVICE COLUMNS. Perhaps blinding factors were not taken into account. The max non-poisoned rows is %d" % max_rows)
        assert len(ctx.selector) == len(ctx.advice)
        for i, (advice, q) in enumerate(zip(ctx.advice, ctx.selector)):
            cell = region.assign_advice(gate_columns[gate_index], row_offset, advice)
            key = ContextCell(ctx.type_id, ctx.context_id, i)
            old = copy_manager.assigned_advices.get(key)
            copy_manager.assigned_advices[key] = cell
            assert old is None or old == cell, "Trying to overwrite virtual cell with a different raw cell"
            if (q and row_offset + rotations > max_rows) or row_offset >= max_rows - 1:
                break_points.append(row_offset)
                row_offset = 0
                gate_index += 1
                if rotations > 1 and i + 2 >= rotations:
                    for delta in range(1, rotations - 1):
                        assert not ctx.selector[i - delta], "We do not support overlaps with delta = %d" % delta
                if gate_index >= len(gate_columns):
                    raise RuntimeError("NOT ENOUGH ADVICE COLUMNS. Perhaps blinding factors were not taken into account. The max non-poisoned rows is %d" % max_rows)
                ncell = region.assign_advice(gate_columns[gate_index], row_offset, advice)
                region.constrain_equal(ncell, cell)
            if q:
                region.enable("q_enable", gate_index, row_offset)
            row_offset += 1
    return break_points


def assign_witnesses(threads: Sequence[Context], gate_columns: Sequence[int], region: Region, break_points: Sequence[int]):
    """single_phase.rs:273-312.  Proving-side layout: no constraints, the columns are left exactly at the recorded break points."""
    if not gate_columns:
        assert sum(len(c.advice) for c in threads) == 0, "Trying to assign threads in a phase with no columns"
        return
    bps = iter(break_points)
    break_point = next(bps, None)
    gate_index, row_offset = 0, 0
    for ctx in threads:
        for advice in ctx.advice:
            region.assign_advice(gate_columns[gate_index], row_offset, advice)
            if break_point == row_offset:
                break_point = next(bps, None)
                row_offset = 0
                gate_index += 1
                region.assign_advice(gate_columns[gate_index], row_offset, advice)
            row_offset += 1


class LookupAnyManager:
    """virtual_region/lookups.rs (ADVICE_COLS = 1: the range lookup): cells queued per context tag, iterated in the BTreeMap's tag order"""

    def __init__(self, witness_gen_only: bool, copy_manager: CopyConstraintManager):
        self.witness_gen_only, self.copy_manager = witness_gen_only, copy_manager
        self.cells_to_lookup: Dict[Tuple[str, int], List[AssignedValue]] = {}

    def add_lookup(self, tag, cell: AssignedValue):
        self.cells_to_lookup.setdefault(tag, []).append(cell)

    def _flat(self):
        return [c for tag in sorted(self.cells_to_lookup) for c in self.cells_to_lookup[tag]]

    def total_rows(self) -> int:
        return len(self._flat())

    def assign_raw(self, lookup_columns: Sequence[int], region: Region):
        lookup_offset, lookup_col = 0, 0
        for advice in self._flat():
            if lookup_col >= len(lookup_columns):
                lookup_col = 0
                lookup_offset += 1
            bcell = region.assign_advice(lookup_columns[lookup_col], lookup_offset, advice.value)
            if not self.witness_gen_only:
                constrain_virtual_equals_external(region, advice, bcell, self.copy_manager)
            lookup_col += 1


def constrain_virtual_equals_external(region: Region, virtual_cell: AssignedValue, external_cell: RawCell, copy_manager: CopyConstraintManager):
    """utils/halo2.rs:56-75"""
    acell = copy_manager.assigned_advices.get(virtual_cell.cell)
    if acell is not None:
        region.constrain_equal(acell, external_cell)
    else:
        assert virtual_cell.cell.type_id == EXTERNAL_CELL_TYPE_ID
        copy_manager.assigned_advices[virtual_cell.cell] = external_cell


def copy_manager_assign_raw(copy_manager: CopyConstraintManager, constant_columns: Sequence[int], region: Region):
    """copy_constraints.rs:120-173: constants sorted by (value, cell) and assigned left to right, then top to bottom; advice equalities
    sorted; every equality imposed on the raw cells"""
    copy_manager.constant_equalities.sort(key=lambda t: (t[0], t[1]))
    fixed_col, fixed_offset = 0, 0
    for c, _ in copy_manager.constant_equalities:
        if c not in copy_manager.assigned_constants:
            assert constant_columns, "constants used but the circuit has no constants column"
            copy_manager.assigned_constants[c] = region.assign_fixed(constant_columns[fixed_col], fixed_offset, c)
            fixed_col += 1
            if fixed_col >= len(constant_columns):
                fixed_col = 0
                fixed_offset += 1
    copy_manager.advice_equalities.sort()
    for left, right in copy_manager.advice_equalities:
        region.constrain_equal(copy_manager.assigned_advices[left], copy_manager.assigned_advices[right])
    for c, right in copy_manager.constant_equalities:
        region.constrain_equal(copy_manager.assigned_constants[c], copy_manager.assigned_advices[right])
    copy_manager.assigned_constants.clear()   # keygen_vk and keygen_pk both call this


# ---------------------------------------------------------------------------------------------------------------- the builder
class BaseCircuitBuilder:
    """gates/circuit/builder.rs: first-phase threads + the range lookup manager + assigned instances, and `synthesize` (gates/circuit/mod.rs:159-203)
    over this repository's column numbering (a `shape`: the attributes halo2_lib_amd.testing.build_circuit reads — gate_advice, lookup_advice,
    table_col, constant_cols, q_lookup_col, q_enable_cols, num_fixed_total, num_instance, usable_rows, n, lookup_bits)."""

    def __init__(self, witness_gen_only: bool = False):
        self.witness_gen_only = witness_gen_only
        self.copy_manager = CopyConstraintManager()
        self.threads: List[Context] = []
        self.lookup_manager = LookupAnyManager(witness_gen_only, self.copy_manager)
        self.assigned_instances: List[List[AssignedValue]] = []
        self.break_points: Optional[List[int]] = None

    def new_thread(self) -> Context:
        ctx = Context(self.witness_gen_only, FIRST_PHASE_TYPE_ID, len(self.threads), self.copy_manager)
        self.threads.append(ctx)
        return ctx

    def main(self) -> Context:
        return self.threads[-1] if self.threads else self.new_thread()

    def statistics(self):
        """(total advice cells, distinct constants, cells to look up) — builder.rs:245-258"""
        return (sum(len(c.advice) for c in self.threads), len({c for c, _ in self.copy_manager.constant_equalities}), self.lookup_manager.total_rows())

    def calculate_params(self, k: int, minimum_rows: int, lookup_bits: Optional[int], num_instance_columns: int = 0):
        """builder.rs:260-288 / multi_phase.rs:131-153 -> (k, num_advice, num_lookup_advice, num_fixed, num_instance, lookup_bits)"""
        max_rows = (1 << k) - minimum_rows
        total_advice, total_fixed, total_lookup = self.statistics()
        num_advice = -(-total_advice // max_rows)
        num_fixed = (total_fixed + (1 << k) - 1) >> k
        num_lookup_advice = -(-total_lookup // max_rows)
        return (k, num_advice, num_lookup_advice, num_fixed, num_instance_columns, lookup_bits)

    def synthesize(self, shape):
        """-> (advice, fixed, copies, instances, break_points) with Montgomery-limb columns, as plonk.keygen / plonk.create_proof take them.
        Keygen stage (witness_gen_only False): assign_with_constraints, selectors, constants, copies.  Prover stage: assign_witnesses from
        `self.break_points`; fixed / copies come out empty (the proving key has them)."""
        # gate.max_rows = 2^k - meta.minimum_rows() (flex_gate/mod.rs:142, range/mod.rs:117): minimum_rows = blinding_factors + 3, i.e. two rows
        # fewer than the prover's own usable rows (n - blinding_factors - 1)
        n, max_rows = shape.n, shape.n - (shape.blinding_factors + 3)
        region = Region(n, len(shape.gate_advice) + len(shape.lookup_advice))
        gate_cols, lookup_cols = list(shape.gate_advice), list(shape.lookup_advice)
        if self.witness_gen_only:
            assert self.break_points is not None, "break points not set"
            assign_witnesses(self.threads, gate_cols, region, self.break_points)
        else:
            self.break_points = assign_with_constraints(self.threads, gate_cols, region, self.copy_manager, max_rows)
        # builder.rs:327-375
        if self.lookup_manager.total_rows():
            assert shape.table_col is not None, "range lookups were queued but the circuit was configured without a RangeConfig"
            if shape.q_lookup_col is not None:
                assert len(gate_cols) == 1
                if not self.witness_gen_only:
                    for advice in self.lookup_manager._flat():
                        (kind, col), row = self.copy_manager.assigned_advices[advice.cell]
                        assert row < max_rows, "range lookup assigned to an unusable row"
                        assert (kind, col) == ("advice", gate_cols[0]), "lookup column does not match"
                        region.enable("q_lookup", 0, row)
            else:
                assert lookup_cols, "range lookups require lookup advice columns"
                assert -(-self.lookup_manager.total_rows() // len(lookup_cols)) <= max_rows, "range lookups would be assigned to unusable rows"
                self.lookup_manager.assign_raw(lookup_cols, region)
        if not self.witness_gen_only:
            copy_manager_assign_raw(self.copy_manager, list(shape.constant_cols), region)
            # builder.rs:291-309
            assert len(self.assigned_instances) == shape.num_instance
            for col, instances in enumerate(self.assigned_instances):
                for i, inst in enumerate(instances):
                    region.constrain_equal(self.copy_manager.assigned_advices[inst.cell], (("instance", col), i))
        return self._export(shape, region)

    def _export(self, shape, region: Region):
        n = shape.n
        advice = [_column(n, col) for col in region.advice]
        fixed, copies = [], []
        if not self.witness_gen_only:
            fcols: List[Dict[int, int]] = [dict() for _ in range(shape.num_fixed_total)]
            if shape.table_col is not None:      # range/mod.rs:154-170
                for v in range(1 << shape.lookup_bits):
                    fcols[shape.table_col][v] = v
            for c, rows in region.fixed.items():
                fcols[c].update(rows)
            for gi, rows in region.selectors.get("q_enable", {}).items():
                for r in rows:
                    fcols[shape.q_enable_cols[gi]][r] = 1
            for rows in region.selectors.get("q_lookup", {}).values():
                for r in rows:
                    fcols[shape.q_lookup_col][r] = 1
            fixed = [_column(n, c) for c in fcols]
            copies = list(region.copies)
        instances = [_limbs([v.value for v in col]) for col in self.assigned_instances]
        return advice, fixed, copies, instances, list(self.break_points or [])


def _limbs(vals) -> np.ndarray:
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = (int(v) << 256) % R_MOD
        out[i] = [(m >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
    return out


def _column(n: int, cells: Dict[int, int]) -> np.ndarray:
    out = np.zeros((n, 4), dtype=np.uint64)
    if cells:
        rows = sorted(cells)
        out[rows] = _limbs([cells[r] for r in rows])
    return out


# ---------------------------------------------------------------------------------------------------------------- a tiny GateChip / RangeChip
class GateChip:
    """the few GateInstructions the layout tests need, with the reference's cell order (flex_gate/mod.rs:158-168, 246-277, 1149-1190)"""

    @staticmethod
    def add(ctx: Context, a, b) -> AssignedValue:
        a, b = _q(a), _q(b)
        return ctx.assign_region_last([a, b, Constant(1), Witness((a.value + b.value) % R_MOD)], [0])

    @staticmethod
    def mul(ctx: Context, a, b) -> AssignedValue:
        a, b = _q(a), _q(b)
        return ctx.assign_region_last([Constant(0), a, b, Witness(a.value * b.value % R_MOD)], [0])

    @staticmethod
    def mul_add(ctx: Context, a, b, c) -> AssignedValue:
        a, b, c = _q(a), _q(b), _q(c)
        return ctx.assign_region_last([c, a, b, Witness((a.value * b.value + c.value) % R_MOD)], [0])

    @staticmethod
    def inner_product(ctx: Context, a: Sequence, b: Sequence) -> AssignedValue:
        """inner_product_simple: | 0 | a0 | b0 | s1 | a1 | b1 | s2 | ... with a gate every three rows — consecutive gates OVERLAP in the
        running-sum cell, the case the break-point copy exists for"""
        a, b = [_q(v) for v in a], [_q(v) for v in b]
        assert len(a) == len(b) and a
        cells, s = [Constant(0)], 0
        for x, y in zip(a, b):
            s = (s + x.value * y.value) % R_MOD
            cells += [x, y, Witness(s)]
        ctx.assign_region(cells, [3 * i for i in range(len(a))])
        return ctx.last()


class RangeChip:
    """range_check of a cell that is already < 2^lookup_bits: queue it for the lookup (range/mod.rs:487-496)"""

    def __init__(self, lookup_bits: int, lookup_manager: LookupAnyManager):
        self.lookup_bits, self.lookup_manager = lookup_bits, lookup_manager

    def add_cell_to_lookup(self, ctx: Context, a: AssignedValue):
        assert a.value < (1 << self.lookup_bits)
        self.lookup_manager.add_lookup(ctx.tag(), a)


def _q(v):
    if isinstance(v, AssignedValue):
        return Existing(v)
    if isinstance(v, int):
        return Witness(v)
    return v
