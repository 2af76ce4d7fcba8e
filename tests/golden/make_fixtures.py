"""Writes tests/golden/reference_vectors.json.

These vectors are HAND-TRANSCRIBED integers from the reference's own unit tests (no reference code is executed or copied; the Rust
reference cannot be built in this environment). Each entry cites the reference file:line it was read from. Running this script just
re-serialises the literals below, so the fixture file and its provenance stay together.
"""
import json
import os

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


vectors = {
    "compile": [  # crates/brainfuck_vm/src/compiler.rs:62-79
        {"code": "++>,<[>+.<-]", "expected": [43, 43, 62, 44, 60, 91, 13, 62, 43, 46, 60, 45, 93, 7]},
    ],
    "trace": [
        {"code": "++", "input": [], "cite": "crates/brainfuck_vm/src/machine.rs:394-431",
         "expected": [[0, 0, 43, 43, 0, 0, 0], [1, 1, 43, 0, 0, 1, 1], [2, 2, 0, 0, 0, 2, INV2]]},
        {"code": "+>,<[>+.<-]", "input": [1], "cite": "crates/brainfuck_prover/src/components/processor/table.rs:698-818", "expected": TRACE_A},
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
    ],
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
