"""Writes tests/golden/reference_vectors.json.

These vectors are HAND-TRANSCRIBED integers from the reference's own unit tests (no reference code is executed or copied; the Rust
reference cannot be built in this environment). Each entry cites the reference file:line it was read from. Running this script just
re-serialises the literals below, so the fixture file and its provenance stay together.
"""
import json
import os

P31 = (1 << 31) - 1  # stwo_prover::core::fields::m31::P (machine.rs:245)
INV2 = 1073741824  # BaseField::from(2).inverse() = 2^-1 mod (2^31 - 1)   (machine.rs:427, processor/table.rs:773)

# Trace of "+>,<[>+.<-]" with input [1]: rows (clk, ip, ci, ni, mp, mv, mvi) — crates/brainfuck_prover/src/components/processor/table.rs:698-818
TRACE_A = [
    [0, 0, 43, 62, 0, 0, 0], [1, 1, 62, 44, 0, 1, 1], [2, 2, 44, 60, 1, 0, 0], [3, 3, 60, 91, 1, 1, 1], [4, 4, 91, 12, 0, 1, 1],
    [5, 6, 62, 43, 0, 1, 1], [6, 7, 43, 46, 1, 1, 1], [7, 8, 46, 60, 1, 2, INV2], [8, 9, 60, 45, 1, 2, INV2], [9, 10, 45, 93, 0, 1, 1],
    [10, 11, 93, 6, 0, 0, 0], [11, 13, 0, 0, 0, 0, 0],
]
# processor table rows: 8 entry fields (clk, ip, ci, ni, mp, mv, mvi, d) + next_clk; padding dummies keep ip, clk increments — processor/table.rs:820-880
PROC_A = [r + [0, TRACE_A[i + 1][0] if i + 1 < len(TRACE_A) else 12] for i, r in enumerate(TRACE_A)]
PROC_A += [[12 + k, 13, 0, 0, 0, 0, 0, 1, 13 + k] for k in range(4)]

# instruction/table.rs:611-744: entries (ip, ci, ni) of "+>,<[>+.<-]" with input [1]; executed instructions appear twice (program + trace)
_A = [(0, 43, 62), (1, 62, 44), (2, 44, 60), (3, 60, 91), (4, 91, 12)]
INS_A = [list(e) + [0] for e in _A for _ in range(2)] + [[5, 12, 62, 0]] + \
        [list(e) + [0] for e in [(6, 62, 43), (7, 43, 46), (8, 46, 60), (9, 60, 45), (10, 45, 93), (11, 93, 6)] for _ in range(2)] + \
        [[12, 6, 0, 0], [13, 0, 0, 0]] + [[13, 0, 0, 1]] * 7
# instruction/table.rs:746-804: "[-]" with no input — only the first instruction is executed
INS_B = [[0, 91, 4, 0], [0, 91, 4, 0], [1, 4, 45, 0], [2, 45, 93, 0], [3, 93, 2, 0], [4, 2, 0, 0], [5, 0, 0, 0], [5, 0, 0, 1]]


def pair_rows(entries):
    """InstructionTable::from(intermediate) — instruction/table.rs:116-145."""
    nxt = entries[1:] + [[entries[-1][0], 0, 0, 1]]
    return [a + b for a, b in zip(entries, nxt)]



# ---- round 3: the remaining table-level tests of the reference, as (register rows -> table rows) vectors ---------------------------------
# A reference test that builds an *intermediate* table from hand-written entries is restated as the register rows that produce exactly
# those entries (Memory/Processor entries are the registers themselves; sub-component entries are selected by `ci`; Instruction entries are
# program + trace, here with an empty program), and the pinned intermediate table is carried to table rows by the cited pairing rule.
def reg(clk=0, ip=0, ci=0, ni=0, mp=0, mv=0, mvi=0):
    return [clk, ip, ci, ni, mp, mv, mvi]


def mem_rows(entries):
    """MemoryTable::from(intermediate) — memory/table.rs:121-151: row r = entry r || entry r+1, the last one pairs with
    new_dummy(last.clk + 1, last.mp, last.mv)."""
    last = entries[-1]
    nxt = entries[1:] + [[last[0] + 1, last[1], last[2], 1]]
    return [a + b for a, b in zip(entries, nxt)]


LEFT, RIGHT, PLUS, MINUS, PUTC, READC, JZ, JNZ = 60, 62, 43, 45, 46, 44, 91, 93

TABLES_R3 = [
    # ---------------- memory/table.rs ----------------
    {"cite": "crates/brainfuck_prover/src/components/memory/table.rs:638-651 (test_sort: sorted by (mp, clk)) + pad :291-303 + pairing :121-151",
     "component": 0, "code_words": [PLUS],
     "trace": [reg(clk=0, mp=1), reg(clk=0, mp=0), reg(clk=1, mp=0)],
     "expected": mem_rows([[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [1, 1, 0, 1]])},
    {"cite": "crates/brainfuck_prover/src/components/memory/table.rs:663-686 (test_complete_wih_dummy_entries: clk gaps of one mp filled with d = 1) + pad + pairing",
     "component": 0, "code_words": [PLUS],
     "trace": [reg(clk=5, mp=1, mv=1), reg(clk=0, mp=0), reg(clk=0, mp=1)],
     "expected": mem_rows([[0, 0, 0, 0], [0, 1, 0, 0], [1, 1, 0, 1], [2, 1, 0, 1], [3, 1, 0, 1], [4, 1, 0, 1], [5, 1, 1, 0], [6, 1, 1, 1]])},
    {"cite": "crates/brainfuck_prover/src/components/memory/table.rs:695-712 (test_pad: dummy (last.clk + 1, last.mp, last.mv)) + pairing",
     "component": 0, "code_words": [PLUS],
     "trace": [reg(clk=0, mp=0, mv=0), reg(clk=1, mp=1, mv=0), reg(clk=2, mp=1, mv=1)],
     "expected": mem_rows([[0, 0, 0, 0], [1, 1, 0, 0], [2, 1, 1, 0], [3, 1, 1, 1]])},
    {"cite": "crates/brainfuck_prover/src/components/memory/table.rs:757-799 (test_trace_evaluation: columns clk, mp, mv, d of rows 0, 1) + pairing",
     "component": 0, "code_words": [PLUS],
     "trace": [reg(clk=0, mp=43, mv=91), reg(clk=1, mp=91, mv=9)],
     "expected": mem_rows([[0, 43, 91, 0], [1, 91, 9, 0]])},
    {"cite": "crates/brainfuck_prover/src/components/memory/table.rs:886-929 (test_interaction_trace_evaluation_dummy_entries_effect: the table of the real entries only)",
     "component": 0, "code_words": [PLUS],
     "trace": [reg(clk=0, mp=43, mv=91), reg(clk=2, mp=91, mv=9)],
     "expected": mem_rows([[0, 43, 91, 0], [2, 91, 9, 0]])},
    # ---------------- instruction/table.rs (entries given directly = trace rows with an empty program) ----------------
    {"cite": "crates/brainfuck_prover/src/components/instruction/table.rs:816-889 (test_trace_evaluation_single_row: all 8 columns)",
     "component": 1, "code_words": [],
     "trace": [reg(ip=1, ci=43, ni=91)],
     "expected": [[1, 43, 91, 0, 1, 0, 0, 1]]},
    {"cite": "crates/brainfuck_prover/src/components/instruction/table.rs:892-946 (test_instruction_trace_evaluation: columns ip, ci, ni) + pairing :116-145",
     "component": 1, "code_words": [],
     "trace": [reg(clk=0, ip=0, ci=43, ni=91), reg(clk=1, ip=1, ci=91, ni=9)],
     "expected": [[0, 43, 91, 0, 1, 91, 9, 0], [1, 91, 9, 0, 1, 0, 0, 1]]},
    # ---------------- processor/table.rs ----------------
    {"cite": "crates/brainfuck_prover/src/components/processor/table.rs:898-937 (test_trace_evaluation_single_row_processor_table) + pairing :117-145",
     "component": 3, "code_words": [PLUS],
     "trace": [reg(1, 2, 3, 4, 5, 6, 7)],
     "expected": [[1, 2, 3, 4, 5, 6, 7, 0, 2]]},
    {"cite": "crates/brainfuck_prover/src/components/processor/table.rs:940-1050 (test_trace_evaluation_processor_table_with_multiple_rows: all 9 columns)",
     "component": 3, "code_words": [PLUS],
     "trace": [reg(0, 1, 2, 3, 4, 5, 6), reg(1, 2, 3, 4, 5, 6, 7)],
     "expected": [[0, 1, 2, 3, 4, 5, 6, 0, 1], [1, 2, 3, 4, 5, 6, 7, 0, 2]]},
    # ---------------- processor/instructions/table.rs (Left) ----------------
    {"cite": "crates/brainfuck_prover/src/components/processor/instructions/table.rs:790-922 (test_trace_evaluation_processor_instruction_table_with_multiple_rows)",
     "component": 7, "code_words": [PLUS],
     "trace": [reg(0, 0, LEFT, PLUS, 4, 5, 6), reg(1, 1, PLUS, LEFT, 1, 2, 3), reg(2, 2, LEFT, MINUS, 4, 5, 6), reg(3, 3, MINUS, 0, 1, 2, 3)],
     # (clk, ip, ci, ni, mp, mv, mvi, d, next_ip, next_mp, next_mv)
     "expected": [[0, 0, LEFT, PLUS, 4, 5, 6, 0, 1, 1, 2], [2, 2, LEFT, MINUS, 4, 5, 6, 0, 3, 1, 2]]},
    # ---------------- processor/instructions/jump/table.rs ----------------
    {"cite": "crates/brainfuck_prover/src/components/processor/instructions/jump/table.rs:759-822 (test_trace_evaluation_single_row_jump_table)",
     "component": 4, "code_words": [PLUS],
     "trace": [reg(1, 2, JNZ, 1, 5, 0, 0), reg(2, 3, 0, 0, 5, 0, 0)],
     # (clk, ip, ci, ni, mp, mv, mvi, next_clk, next_ip, next_mp, next_mv, d, is_mv_zero)
     "expected": [[1, 2, JNZ, 1, 5, 0, 0, 2, 3, 5, 0, 0, 1]]},
    {"cite": "crates/brainfuck_prover/src/components/processor/instructions/jump/table.rs:825-978 (test_trace_evaluation_jump_table_with_multiple_rows: all 13 columns)",
     "component": 4, "code_words": [PLUS],
     "trace": [reg(11, 12, JNZ, 7, 0, 1, 1), reg(12, 7, RIGHT, PLUS, 0, 1, 1), reg(17, 12, JNZ, 7, 0, 0, 0), reg(18, 14, 0, 0, 0, 0, 0)],
     "expected": [[11, 12, JNZ, 7, 0, 1, 1, 12, 7, 0, 1, 0, 0], [17, 12, JNZ, 7, 0, 0, 0, 18, 14, 0, 0, 0, 1]]},
    # ---------------- end_of_execution/table.rs ----------------
    {"cite": "crates/brainfuck_prover/src/components/processor/instructions/end_of_execution/table.rs:394-427 (test_trace_evaluation_single_row_end_of_execution_table)",
     "component": 12, "code_words": [PLUS],
     "trace": [reg(clk=1, ip=2)],
     "expected": [[1, 2, 0, 0, 0, 0, 0]]},
]

# The constructor-level tests (entry / row / add_entry literals), carried through the same builders: one register row per literal entry.
TABLES_R3 += [
    {"cite": "memory/table.rs:526-537,592-602 (test_memory_entry_new, test_add_entry: entry (0, 43, 91), d = 0) + pairing :121-151 with new_dummy :539-553 (d = 1)",
     "component": 0, "code_words": [PLUS], "trace": [reg(clk=0, mp=43, mv=91)], "expected": mem_rows([[0, 43, 91, 0]])},
    {"cite": "memory/table.rs:623-635 (test_add_multiple_entries literals) through sort :249-251, pad :291-303, pairing",
     "component": 0, "code_words": [PLUS], "trace": [reg(clk=0, mp=43, mv=91), reg(clk=1, mp=91, mv=9), reg(clk=43, mp=62, mv=43)],
     "expected": mem_rows([[0, 43, 91, 0], [43, 62, 43, 0], [1, 91, 9, 0], [2, 91, 9, 1]])},
    {"cite": "instruction/table.rs:511-524,526-538 (test_new_instruction_entry (1, 43, 45) d = 0; new_dummy(ip): ci = ni = 0, d = 1) + pairing :116-145",
     "component": 1, "code_words": [], "trace": [reg(ip=1, ci=43, ni=45)], "expected": [[1, 43, 45, 0, 1, 0, 0, 1]]},
    {"cite": "instruction/table.rs:561-575 (test_add_entry literal (0, 43, 91)) + pairing",
     "component": 1, "code_words": [], "trace": [reg(ip=0, ci=43, ni=91)], "expected": [[0, 43, 91, 0, 0, 0, 0, 1]]},
    {"cite": "instruction/table.rs:577-594 (test_add_multiple_entries literals) through pad :239-248 and pairing",
     "component": 1, "code_words": [], "trace": [reg(clk=0, ip=0, ci=43, ni=91), reg(clk=1, ip=1, ci=91, ni=9), reg(clk=2, ip=2, ci=62, ni=43)],
     "expected": [[0, 43, 91, 0, 1, 91, 9, 0], [1, 91, 9, 0, 2, 62, 43, 0], [2, 62, 43, 0, 2, 0, 0, 1], [2, 0, 0, 1, 2, 0, 0, 1]]},
    {"cite": "program/table.rs:278-292,315-328 (test_row_new / test_table_add_row: row (0, PutChar, 91), d = 0) as the first row of the program [PutChar, 91]",
     "component": 2, "code_words": [PUTC, 91], "trace": [reg()], "expected": [[0, PUTC, 91, 0], [1, 91, 0, 0]]},
    {"cite": "processor/table.rs:544-569 (test_processor_table_entry_from_registers) + row :571-600 with the builder's dummy (last.clk + 1, last.ip)",
     "component": 3, "code_words": [PLUS], "trace": [reg(1, 5, 43, 91, 2, 7, 0)], "expected": [[1, 5, 43, 91, 2, 7, 0, 0, 2]]},
    {"cite": "processor/table.rs:616-633 (test_add_entry literal)",
     "component": 3, "code_words": [PLUS], "trace": [reg(10, 15, 43, 91, 20, 25, 1)], "expected": [[10, 15, 43, 91, 20, 25, 1, 0, 11]]},
    {"cite": "processor/table.rs:635-676 (test_add_multiple_entries literals) through pad :241-253 and pairing :117-145",
     "component": 3, "code_words": [PLUS], "trace": [reg(1, 5, 43, 91, 10, 15, 0), reg(2, 6, 44, 92, 11, 16, 1), reg(3, 7, 45, 93, 12, 17, 0)],
     "expected": [[1, 5, 43, 91, 10, 15, 0, 0, 2], [2, 6, 44, 92, 11, 16, 1, 0, 3], [3, 7, 45, 93, 12, 17, 0, 0, 4], [4, 7, 0, 0, 0, 0, 0, 1, 5]]},
    {"cite": "processor/instructions/table.rs:518-575 (test_processor_instruction_table_entry_from_registers / _row: entry (1, 5, '+', 91, 2, 7, 0) paired with a dummy of the "
             "same ip: next_ip = 5, next_mp = next_mv = 0) — the Plus table, whose opcode the literal carries",
     "component": 10, "code_words": [PLUS], "trace": [reg(1, 5, PLUS, 91, 2, 7, 0), reg(2, 5, 0, 0, 0, 0, 0)],
     "expected": [[1, 5, PLUS, 91, 2, 7, 0, 0, 5, 0, 0]]},
    {"cite": "end_of_execution/table.rs:318-336 (test_add_row literal: (10, 15, 0, 91, 20, 25, 1)) — the row with ci = 0 is the table",
     "component": 12, "code_words": [PLUS], "trace": [reg(9, 14, PLUS, 0, 20, 24, 1), reg(10, 15, 0, 91, 20, 25, 1)],
     "expected": [[10, 15, 0, 91, 20, 25, 1]]},
]

# Error paths of the table builders (TraceError, crates/brainfuck_prover/src/components/mod.rs): what the builders must refuse.
TABLE_ERRORS = [
    {"cite": "memory/table.rs:749-755 (test_empty_trace_evaluation)", "component": 0, "code_words": [PLUS], "trace": [], "error": "EmptyTrace"},
    {"cite": "instruction/table.rs:596-608,806-812 (empty registers and program -> empty table -> EmptyTrace)", "component": 1, "code_words": [], "trace": [], "error": "EmptyTrace"},
    {"cite": "program/table.rs:384-390 (test_trace_evaluation_empty_table)", "component": 2, "code_words": [], "trace": [], "error": "EmptyTrace"},
    {"cite": "processor/table.rs:889-895 (test_trace_evaluation_empty_processor_table)", "component": 3, "code_words": [PLUS], "trace": [], "error": "EmptyTrace"},
    {"cite": "end_of_execution/table.rs:373-378 (no row with ci = 0)", "component": 12, "code_words": [PLUS], "trace": [reg(0, 0, PLUS, 0, 0, 0, 0)], "error": "InvalidEndOfExecution"},
    {"cite": "end_of_execution/table.rs:381-391 (two rows with ci = 0)", "component": 12, "code_words": [PLUS], "trace": [reg(0, 0, 0, 0, 0, 0, 0), reg(1, 1, 0, 0, 0, 0, 0)],
     "error": "InvalidEndOfExecution"},
    {"cite": "jump/table.rs:191-193 (a jump row pairs a real entry with a dummy one: both entries must share d)", "component": 4, "code_words": [PLUS],
     "trace": None, "error": None},     # placeholder removed below: not reachable from a register trace (pairs are pushed together)
]
TABLE_ERRORS = [e for e in TABLE_ERRORS if e["error"]]

# The 10 negative AIR tests of the Memory component (memory/component.rs:211-609): a table built from registers, single cells patched, and the
# row / value stwo's assert_constraints reports ("row: r, left: (v + 0i) + (0 + 0i)u"): row r of the trace domain in natural order = table row r
# (its first SIMD lane), v = the first constraint that does not vanish there. `constraint` = its index in MemoryEval::evaluate's order
# (memory/component.rs:81-121). Main columns: clk, mp, mv, d, next_clk, next_mp, next_mv, next_d.
AIR_NEGATIVE = [
    {"cite": "memory/component.rs:215-252 (test_invalid_boundary_clk)", "component": 0, "trace": [reg(clk=1)], "patch": [], "table_row": 0, "value": 1, "constraint": 0},
    {"cite": "memory/component.rs:255-289 (test_invalid_boundary_mp)", "component": 0, "trace": [reg(mp=1)], "patch": [], "table_row": 0, "value": 1, "constraint": 1},
    {"cite": "memory/component.rs:292-326 (test_invalid_boundary_mv)", "component": 0, "trace": [reg(mv=1)], "patch": [], "table_row": 0, "value": 1, "constraint": 2},
    {"cite": "memory/component.rs:329-365 (test_invalid_boundary_d: table[0].d = 1)", "component": 0, "trace": [reg()], "patch": [[0, 3, 1]], "table_row": 0, "value": 1, "constraint": 3},
    {"cite": "memory/component.rs:368-403 (test_invalid_transition_mp_increase: mp jumps by 2)", "component": 0, "trace": [reg(), reg(mp=2)], "patch": [],
     "table_row": 0, "value": 2, "constraint": 6},
    {"cite": "memory/component.rs:406-441 (test_invalid_transition_clk_increase: same mp, clk not increased)", "component": 0, "trace": [reg(), reg()], "patch": [],
     "table_row": 0, "value": 1, "constraint": 7},
    {"cite": "memory/component.rs:444-482 (test_invalid_transition_mp_increase_next_mv: mp + 1 but next_mv != 0)", "component": 0, "trace": [reg(), reg(mp=1, mv=1)], "patch": [],
     "table_row": 0, "value": 1, "constraint": 8},
    {"cite": "memory/component.rs:485-522 (test_invalid_transition_next_dummy: table[0].next_d = 2)", "component": 0, "trace": [reg(), reg(mp=1)], "patch": [[0, 7, 2]],
     "table_row": 0, "value": 2, "constraint": 5},
    {"cite": "memory/component.rs:525-565 (test_invalid_transition_d_mp: table[1].d = 1, table[1].next_mp = 2)", "component": 0, "trace": [reg(), reg(mp=1)],
     "patch": [[1, 3, 1], [1, 5, 2]], "table_row": 1, "value": 1, "constraint": 9},
    {"cite": "memory/component.rs:570-609 (test_invalid_transition_d_mv: table[1].d = 1, table[1].next_mv = 1)", "component": 0, "trace": [reg(), reg(mp=1)],
     "patch": [[1, 3, 1], [1, 6, 1]], "table_row": 1, "value": 1, "constraint": 10},
]

# logUp structure of the 7 interaction-trace tests: per table row the numerator the reference writes (write_frac) and which main columns
# enter the denominator `combine` (in that order), per logUp column. Evaluated under LookupElements::dummy() (z = 1, all alpha powers 1).
#   rows: explicit main-table rows, or code/input (the table the component builds from the executed program)
LOGUP_STRUCTURE = [
    {"cite": "crates/brainfuck_prover/src/components/memory/table.rs:811-878 (test_interaction_trace_evaluation)",
     "component": 0, "rows": mem_rows([[0, 0, 0, 0], [1, 1, 0, 1], [2, 1, 0, 0], [3, 1, 0, 1]]),
     "columns": [{"denominator_columns": [0, 1, 2], "numerators": [-1, 0, -1, 0]}]},
    {"cite": "crates/brainfuck_prover/src/components/instruction/table.rs:983-1054 (test_interaction_trace_evaluation, program '+->[-]')",
     "component": 1, "code": "+->[-]", "input": [],
     "columns": [{"denominator_columns": [0, 1, 2], "numerators": [-1] * 13 + [0] * 3}]},
    {"cite": "crates/brainfuck_prover/src/components/program/table.rs:587-647 (test_interaction_trace_evaluation, program '+->[-]')",
     "component": 2, "code": "+->[-]", "input": [],
     "columns": [{"denominator_columns": [0, 1, 2], "numerators": [1] * 8}]},
    {"cite": "crates/brainfuck_prover/src/components/processor/table.rs:1070-1178 (test_interaction_trace_evaluation, program '+,.' input 1): three logUp columns in "
             "the order Processor(clk, ip, ci, ni, mp, mv, mvi), Instruction(ip, ci, ni), Memory(clk, mp, mv)",
     "component": 3, "code": "+,.", "input": [1],
     "columns": [{"denominator_columns": [0, 1, 2, 3, 4, 5, 6], "numerators": [1, 1, 1, 1]},
                 {"denominator_columns": [1, 2, 3], "numerators": [1, 1, 1, 1]},
                 {"denominator_columns": [0, 4, 5], "numerators": [1, 1, 1, 1]}]},
    {"cite": "crates/brainfuck_prover/src/components/processor/instructions/table.rs:924-1000 (test_interaction_trace_evaluation, Left table of '+>>><,<.<' input 1)",
     "component": 7, "code": "+>>><,<.<", "input": [1],
     "columns": [{"denominator_columns": [0, 1, 2, 3, 4, 5, 6], "numerators": [-1, -1, -1, 0]}]},
    {"cite": "crates/brainfuck_prover/src/components/processor/instructions/jump/table.rs:980-1052 (test_interaction_trace_evaluation, JumpIfNotZero table of '++>,<[>+.<-]' input 1)",
     "component": 4, "code": "++>,<[>+.<-]", "input": [1],
     "columns": [{"denominator_columns": [0, 1, 2, 3, 4, 5, 6], "numerators": [-1, -1]}]},
    {"cite": "crates/brainfuck_prover/src/components/processor/instructions/end_of_execution/table.rs:431-502 (test_interaction_trace_evaluation, '+>>><,<.<' input 1)",
     "component": 12, "code": "+>>><,<.<", "input": [1],
     "columns": [{"denominator_columns": [0, 1, 2, 3, 4, 5, 6], "numerators": [-1]}]},
]

# The reference's 13 positive AIR tests (`test_*_constraints`): table of one component from a program run, logUp under LookupElements::dummy(),
# `assert_constraints` on CanonicCoset(LOG_SIZE) passes. LOG_SIZE (a literal of each test) pins the table's padded size: 2^LOG_SIZE = 16 * rows.
_C = "crates/brainfuck_prover/src/components/"
AIR_POSITIVE = [
    {"cite": _C + "memory/component.rs:163-212 (test_memory_constraints)", "component": 0, "code": "+>,<[>+.<-]", "input": [1], "log_size": 9},
    {"cite": _C + "instruction/component.rs:164-213 (test_instruction_constraints)", "component": 1, "code": "+>,<[>+.<-]", "input": [1], "log_size": 9},
    {"cite": _C + "program/component.rs:129-176 (test_program_constraints)", "component": 2, "code": "+>,<[>+.<-]", "input": [1], "log_size": 8},
    {"cite": _C + "processor/component.rs:178-230 (test_processor_constraints)", "component": 3, "code": "+++>,<[>+.<-]", "input": [1], "log_size": 9},
    {"cite": _C + "processor/instructions/jump/jump_if_not_zero_component.rs:154-202 (test_jump_if_not_zero_constraints)", "component": 4, "code": "+++>,<[>+.<-]", "input": [1], "log_size": 6},
    {"cite": _C + "processor/instructions/jump/jump_if_zero_component.rs:154-202 (test_jump_if_zero_constraints)", "component": 5, "code": "[][]+[-]", "input": [], "log_size": 6},
    {"cite": _C + "processor/instructions/input_component.rs:142-190 (test_input_instruction_constraints)", "component": 6, "code": "+++>,<[>+.<-]", "input": [1], "log_size": 4},
    {"cite": _C + "processor/instructions/left_component.rs:142-190 (test_left_instruction_constraints)", "component": 7, "code": "+++>,<[>+.<-]", "input": [1], "log_size": 6},
    {"cite": _C + "processor/instructions/minus_component.rs:144-192 (test_minus_instruction_constraints)", "component": 8, "code": "+++>,<[>+.<-]", "input": [1], "log_size": 6},
    {"cite": _C + "processor/instructions/output_component.rs:145-193 (test_output_instruction_constraints)", "component": 9, "code": "+++>,<[>+.<-]", "input": [1], "log_size": 6},
    {"cite": _C + "processor/instructions/plus_component.rs:145-193 (test_plus_instruction_constraints)", "component": 10, "code": "+++>,<[>+.<-]", "input": [1], "log_size": 7},
    {"cite": _C + "processor/instructions/right_component.rs:142-190 (test_right_instruction_constraints)", "component": 11, "code": "+++>,<[>+.<-]", "input": [1], "log_size": 6},
    {"cite": _C + "processor/instructions/end_of_execution/component.rs:114-162 (test_end_of_execution_instruction_constraints)", "component": 12, "code": "++[-]+.", "input": [1], "log_size": 4},
]

vectors = {
    "compile": [  # crates/brainfuck_vm/src/compiler.rs:62-79
        {"code": "++>,<[>+.<-]", "expected": [43, 43, 62, 44, 60, 91, 13, 62, 43, 46, 60, 45, 93, 7]},
    ],
    "trace": [
        {"code": "++", "input": [], "cite": "crates/brainfuck_vm/src/machine.rs:394-431",
         "expected": [[0, 0, 43, 43, 0, 0, 0], [1, 1, 43, 0, 0, 1, 1], [2, 2, 0, 0, 0, 2, INV2]]},
        {"code": "+>,<[>+.<-]", "input": [1], "cite": "crates/brainfuck_prover/src/components/processor/table.rs:698-818", "expected": TRACE_A},
    ],
    # The VM's single-instruction unit tests (crates/brainfuck_vm/src/machine.rs:291-392). Each test hands the machine the program WORDS and
    # asserts a few fields of the final state: `code_words` are those words (they equal what the compiler emits for `code`), `final` the
    # asserted registers of the last trace row, `ram0` the asserted value of memory cell 0 (observable as `mv` of the last row whose mp is 0),
    # `output` the asserted output bytes.
    "vm_unit": [
        {"cite": "machine.rs:291-301 (test_right_instruction)", "code": ">>", "code_words": [62, 62], "input": [], "final": {"mp": 2}},
        {"cite": "machine.rs:303-312 (test_left_instruction)", "code": ">><", "code_words": [62, 62, 60], "input": [], "final": {"mp": 1}},
        {"cite": "machine.rs:314-324 (test_plus_instruction)", "code": "+", "code_words": [43], "input": [], "final": {"mv": 1}, "ram0": 1},
        {"cite": "machine.rs:326-336 (test_minus_instruction)", "code": "--", "code_words": [45, 45], "input": [], "final": {"mv": P31 - 2}, "ram0": P31 - 2},
        {"cite": "machine.rs:338-350 (test_read_write_char)", "code": ",.", "code_words": [44, 46], "input": [97], "output": [97]},
        {"cite": "machine.rs:352-370 (test_skip_loop)", "code": "[-]+", "code_words": [91, 4, 45, 93, 2, 43], "input": [], "final": {"mv": 1}, "ram0": 1},
        {"cite": "machine.rs:372-392 (test_enter_loop)", "code": "+[+>]", "code_words": [43, 91, 6, 43, 62, 93, 3], "input": [], "final": {"mp": 1, "mv": 0}, "ram0": 2},
    ],
    "vm_outputs": [  # crates/brainfuck_vm/tests/integration.rs:12-104
        {"program": "a-bc.bf", "input": [97], "expected": [98, 99]},
        {"program": "collatz.bf", "input": [0x37, 10], "expected": [0x31, 0x36, 10]},
        {"program": "hello1.bf", "input": [], "expected": list(b"Hello World!\n")},
        {"program": "hello2.bf", "input": [], "expected": list(b"Hello World!\n")},
        {"program": "hello3.bf", "input": [], "expected": list(b"Hello, World!\n")},
        {"program": "hello4.bf", "input": [], "expected": list(b"Hello World!\n")},
        {"program": "hello_kakarot.bf", "input": [], "expected": list(b"Hello Kakarot World!\n")},
        {"program": "fib19.bf", "input": [], "expected": [85]},
    ],
    # tables built from explicit register traces / programs; component index = claim order (mod.rs:85-99)
    "tables": [
        {"cite": "crates/brainfuck_prover/src/components/memory/table.rs:714-746 (test_memory_intermediate_table_from_registers) + pairing :121-151",
         "component": 0, "code_words": [43],
         "trace": [[5, 0, 0, 0, 1, 1, 0], [0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 1, 0, 0]],
         # intermediate: (0,0,0,0) (1,1,0,0) (2,1,0,1) (3,1,0,1) (4,1,0,1) (5,1,1,0) (6,1,1,1) (7,1,1,1); extra dummy (8,1,1,1)
         "expected": [[0, 0, 0, 0, 1, 1, 0, 0], [1, 1, 0, 0, 2, 1, 0, 1], [2, 1, 0, 1, 3, 1, 0, 1], [3, 1, 0, 1, 4, 1, 0, 1], [4, 1, 0, 1, 5, 1, 1, 0],
                      [5, 1, 1, 0, 6, 1, 1, 1], [6, 1, 1, 1, 7, 1, 1, 1], [7, 1, 1, 1, 8, 1, 1, 1]]},
        {"cite": "crates/brainfuck_prover/src/components/processor/table.rs:678-887 (test_processor_table_from_registers_example_program)",
         "component": 3, "code": "+>,<[>+.<-]", "input": [1], "expected": PROC_A},
        {"cite": "crates/brainfuck_prover/src/components/processor/instructions/table.rs:653-728 (test_left_table_from_registers_example_program)",
         "component": 7, "code": "+>,<[>+.<-]", "input": [1],
         # (clk, ip, ci, ni, mp, mv, mvi, d, next_ip, next_mp, next_mv)
         "expected": [[3, 3, 60, 91, 1, 1, 1, 0, 4, 0, 1], [8, 9, 60, 45, 1, 2, INV2, 0, 10, 0, 1]]},
        {"cite": "crates/brainfuck_prover/src/components/processor/instructions/jump/table.rs:665-747 (test_jump_if_not_zero_table_from_registers_example_program)",
         "component": 4, "code": "++>,<[>+.<-]", "input": [1],
         # (clk, ip, ci, ni, mp, mv, mvi, next_clk, next_ip, next_mp, next_mv, d, is_mv_zero)
         "expected": [[11, 12, 93, 7, 0, 1, 1, 12, 7, 0, 1, 0, 0], [17, 12, 93, 7, 0, 0, 0, 18, 14, 0, 0, 0, 1]]},
        # Instruction table: the reference test pins the *intermediate* table (ip, ci, ni, d): program and trace merged, sorted by
        # (ip, clk), padded with dummies (d = 1) to a power of two; the table row is (entry r || entry r+1) and the last row pairs
        # with new_dummy(last ip) (instruction/table.rs:116-145). The expected rows below are that pairing applied to the pinned entries.
        {"cite": "crates/brainfuck_prover/src/components/instruction/table.rs:611-744 (test_instruction_intermediate_table_from_registers_example_program) + pairing :116-145",
         "component": 1, "code": "+>,<[>+.<-]", "input": [1], "expected": pair_rows(INS_A)},
        {"cite": "crates/brainfuck_prover/src/components/instruction/table.rs:746-804 (test_instruction_table_program_unused_instruction) + pairing :116-145",
         "component": 1, "code": "[-]", "input": [], "expected": pair_rows(INS_B)},
        {"cite": "crates/brainfuck_prover/src/components/program/table.rs:357-382 (test_program_table_from_program_memory)",
         "component": 2, "code": "+>-", "input": [1],
         # (ip, ci, ni, d); the padding row is new_dummy(last ip)
         "expected": [[0, 43, 62, 0], [1, 62, 45, 0], [2, 45, 0, 0], [2, 0, 0, 1]]},
        {"cite": "crates/brainfuck_prover/src/components/processor/instructions/end_of_execution/table.rs:339-370 (test_end_of_execution_table_from_registers_example_program)",
         "component": 12, "code": "+>,<[>+.<-]", "input": [1],
         # (clk, ip, ci, ni, mp, mv, mvi)
         "expected": [[11, 13, 0, 0, 0, 0, 0]]},
    ] + TABLES_R3,
    "table_errors": TABLE_ERRORS,
    "logup_structure": LOGUP_STRUCTURE,
    "air_negative": AIR_NEGATIVE,
    "air_positive": AIR_POSITIVE,
    # component log sizes measured for the bundled programs — SURVEY.md Appendix A.3 (derived from the reference's padding rules)
    "log_sizes": [
        {"program": "hello_kakarot.bf", "input": [], "steps": 651, "expected": [17, 14, 12, 14, 8, 4, 4, 10, 10, 9, 13, 11, 4]},
        {"program": "fib19.bf", "input": [], "steps": 199246, "expected": [24, 22, 11, 22, 19, 11, 4, 20, 19, 4, 20, 20, 4]},
        {"program": "a-bc.bf", "input": [97], "steps": 19, "expected": [9, 10, 8, 9, 5, 4, 4, 6, 5, 5, 6, 6, 4]},
        {"program": "loop.bf", "input": [], "steps": 2, "expected": [5, 8, 7, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4]},
    ],
    # public known-answer tests: RFC 7693 Appendix B ("abc") and the empty string for BLAKE2s-256
    "blake2s": [
        {"msg_hex": "616263", "digest_hex": "508c5e8c327c14e2e1a72ba34eeb452f37458b209ed63a294d999b4c86675982"},
        {"msg_hex": "", "digest_hex": "69217a3079908094e11121d042354a7c1f55b6482ca1a51e1b250dfd1ed0eef9"},
    ],
}

if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
    json.dump(vectors, open(path, "w"), indent=1)
    print("wrote", path)
