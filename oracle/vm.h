// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header).
//
// The Brainfuck compiler and virtual machine, stated the way the reference states them — one small object per reference type, field
// values kept as M31 and advanced with field arithmetic, one method per reference method — and deliberately NOT the way the product's host
// side (stwo-brainfuck_amd/csrc/host/vm.h: flat u32 registers, a switch, a memoised inverse) is written, so that the differential fuzzer
// (tools/fuzz_vm.py) compares two texts that share a specification and nothing else.
//
//   Compiler            crates/brainfuck_vm/src/compiler.rs:6-38
//   InstructionKind     crates/brainfuck_vm/src/instruction.rs:63-94   (TryFrom<u8>: anything else is an invalid instruction)
//   ProgramMemory, MutableState, Machine
//                       crates/brainfuck_vm/src/machine.rs:31-36, 92-98, 141-238
//
// Pinned by the reference's own vectors (tests/golden/reference_vectors.json): compiler.rs:62-79, machine.rs:291-431 (the seven
// single-instruction cases and the "++" trace), crates/brainfuck_vm/tests/integration.rs:12-104 (program outputs, fib19 -> [85]).
#pragma once
#include "field.h"
#include <string>
#include <stdexcept>

namespace orc {

// crates/brainfuck_vm/src/registers.rs:6-21 — the row type every other oracle file consumes (canonical u32 representatives).
struct Registers { u32 clk = 0, ip = 0, ci = 0, ni = 0, mp = 0, mv = 0, mvi = 0; };

// crates/brainfuck_vm/src/instruction.rs:65-76 — the opcodes are the ASCII codes of the eight symbols.
enum : u32 { OP_RIGHT = '>', OP_LEFT = '<', OP_PLUS = '+', OP_MINUS = '-', OP_PUTCHAR = '.', OP_READCHAR = ',', OP_JZ = '[', OP_JNZ = ']' };

// ---- Compiler (compiler.rs) -------------------------------------------------------------------------------------------------------
class Compiler {
    std::vector<char> symbols_;          // `code`: the program text without whitespace (compiler.rs:13-15)
    std::vector<M31> instructions_;

    // char::is_whitespace restricted to what a byte string can hold: the ASCII white-space set (U+0009..U+000D, U+0020)
    static bool is_whitespace(unsigned char c) { return c == 0x20 || (c >= 0x09 && c <= 0x0D); }

   public:
    explicit Compiler(const std::string& text) {
        for (std::string::size_type k = 0; k < text.size(); ++k)
            if (!is_whitespace(static_cast<unsigned char>(text[k]))) symbols_.push_back(text[k]);
    }

    // compiler.rs:17-37
    std::vector<M31> compile() {
        std::vector<std::size_t> loop_stack;
        for (char symbol : symbols_) {
            instructions_.push_back(M31::from(static_cast<unsigned char>(symbol)));
            if (symbol == '[') {
                instructions_.push_back(M31(0));                         // placeholder for the jump target
                loop_stack.push_back(instructions_.size() - 1);
            } else if (symbol == ']') {
                if (loop_stack.empty()) throw std::runtime_error("unbalanced ]");   // `loop_stack.pop().unwrap()`
                const std::size_t start_pos = loop_stack.back();
                loop_stack.pop_back();
                instructions_[start_pos] = M31::from(instructions_.size());
                instructions_.push_back(M31::from(start_pos + 1));
            }
        }
        return instructions_;
    }
};

// the form the rest of the oracle (and its C API) uses: program words as canonical u32
static inline std::vector<u32> compile(const std::string& code) {
    Compiler compiler(code);
    std::vector<u32> words;
    for (const M31& w : compiler.compile()) words.push_back(w.v);
    return words;
}

// ---- InstructionType::try_from(u8) (instruction.rs:78-94) ------------------------------------------------------------------------------
enum class InstructionKind { Right, Left, Plus, Minus, PutChar, ReadChar, JumpIfZero, JumpIfNotZero };
static inline InstructionKind instruction_kind_from(u32 value) {
    const unsigned char byte = static_cast<unsigned char>(value);       // `self.state.registers.ci.0 as u8`
    if (byte == '>') return InstructionKind::Right;
    if (byte == '<') return InstructionKind::Left;
    if (byte == '+') return InstructionKind::Plus;
    if (byte == '-') return InstructionKind::Minus;
    if (byte == '.') return InstructionKind::PutChar;
    if (byte == ',') return InstructionKind::ReadChar;
    if (byte == '[') return InstructionKind::JumpIfZero;
    if (byte == ']') return InstructionKind::JumpIfNotZero;
    throw std::runtime_error("invalid instruction");
}

// ---- Machine (machine.rs) ------------------------------------------------------------------------------------------------------------
class Machine {
    struct FieldRegisters { M31 clk, ip, ci, ni, mp, mv, mvi; };          // registers.rs, as field elements
    struct ProgramMemory { std::vector<M31> code; };                      // machine.rs:31-36
    struct MutableState { std::vector<M31> ram; FieldRegisters registers; };   // machine.rs:92-98

    ProgramMemory program_;
    MutableState state_;
    std::vector<u8> input_;
    std::size_t input_cursor_ = 0;

    M31& cell_under_pointer() { return state_.ram.at(state_.registers.mp.v); }   // an out-of-range pointer is an error here, a panic there

    // machine.rs:236-238
    void write_trace() {
        const FieldRegisters& r = state_.registers;
        Registers row;
        row.clk = r.clk.v; row.ip = r.ip.v; row.ci = r.ci.v; row.ni = r.ni.v; row.mp = r.mp.v; row.mv = r.mv.v; row.mvi = r.mvi.v;
        trace.push_back(row);
    }
    // machine.rs:231-234
    void next_clock_cycle() {
        state_.registers.clk += M31(1);
        state_.registers.ip += M31(1);
    }
    // machine.rs:163-169 — `read_exact` of one byte: an exhausted input is an I/O error
    void read_char() {
        if (input_cursor_ == input_.size()) throw std::runtime_error("input exhausted");
        cell_under_pointer() = M31::from(input_[input_cursor_]);
        input_cursor_ += 1;
    }
    // machine.rs:171-175
    void write_char() { output.push_back(static_cast<u8>(cell_under_pointer().v)); }

    // machine.rs:177-229
    void execute_instruction(InstructionKind ins) {
        FieldRegisters& reg = state_.registers;
        if (ins == InstructionKind::Right) {
            reg.mp += M31(1);
        } else if (ins == InstructionKind::Left) {
            reg.mp -= M31(1);
        } else if (ins == InstructionKind::Plus) {
            cell_under_pointer() += M31(1);
        } else if (ins == InstructionKind::Minus) {
            cell_under_pointer() -= M31(1);
        } else if (ins == InstructionKind::ReadChar) {
            read_char();
        } else if (ins == InstructionKind::PutChar) {
            write_char();
        } else if (ins == InstructionKind::JumpIfZero) {
            const M31 argument = program_.code.at((reg.ip + M31(1)).v);
            reg.ni = argument;
            if (cell_under_pointer().is_zero()) {
                reg.ip = argument;
                return;                                                   // mv and mvi keep their values (machine.rs:205-208)
            }
            reg.ip += M31(1);
        } else {                                                          // JumpIfNotZero
            const M31 argument = program_.code.at((reg.ip + M31(1)).v);
            if (!cell_under_pointer().is_zero()) {
                reg.ip = argument - M31(1);
                return;
            }
            reg.ip += M31(1);
        }
        reg.mv = cell_under_pointer();
        reg.mvi = reg.mv.is_zero() ? M31(0) : inv(reg.mv);
    }

   public:
    static constexpr std::size_t DEFAULT_RAM_SIZE = 30000;                // machine.rs:114
    std::vector<u8> output;
    std::vector<Registers> trace;

    Machine(const std::vector<u32>& code, std::vector<u8> input, std::size_t ram_size = DEFAULT_RAM_SIZE) : input_(std::move(input)) {
        for (u32 w : code) program_.code.push_back(M31::from(w));
        state_.ram.assign(ram_size, M31(0));
    }

    // machine.rs:141-161
    void execute() {
        const std::size_t len = program_.code.size();
        while (state_.registers.ip.v < len) {                             // `ip < BaseField::from(code.len())` on canonical values
            FieldRegisters& reg = state_.registers;
            reg.ci = program_.code[reg.ip.v];
            reg.ni = (reg.ip.v == len - 1) ? M31(0) : program_.code[(reg.ip + M31(1)).v];
            write_trace();
            const InstructionKind kind = instruction_kind_from(reg.ci.v);
            execute_instruction(kind);
            next_clock_cycle();
        }
        // last clock cycle
        state_.registers.ci = M31(0);
        state_.registers.ni = M31(0);
        write_trace();
    }
};

}  // namespace orc
