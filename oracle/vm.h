// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header).
// Brainfuck compiler + VM, restating crates/brainfuck_vm/src/compiler.rs:17-37 and crates/brainfuck_vm/src/machine.rs:141-238.
// Pinned by the reference's own golden vectors: compiler.rs:62-79 (compile), machine.rs:394-431 (trace of "++"),
// crates/brainfuck_vm/tests/integration.rs:12-104 (program outputs, fib19 -> [85]).
#pragma once
#include "field.h"
#include <string>
#include <stdexcept>

namespace orc {

// crates/brainfuck_vm/src/registers.rs:6-21
struct Registers { u32 clk = 0, ip = 0, ci = 0, ni = 0, mp = 0, mv = 0, mvi = 0; };

// crates/brainfuck_vm/src/instruction.rs:65-76 — opcodes are the ASCII codes.
enum : u32 { OP_RIGHT = '>', OP_LEFT = '<', OP_PLUS = '+', OP_MINUS = '-', OP_PUTCHAR = '.', OP_READCHAR = ',', OP_JZ = '[', OP_JNZ = ']' };

// compiler.rs:13-37: strip whitespace; each symbol -> its code; '[' and ']' are followed by a jump-target word.
static inline std::vector<u32> compile(const std::string& code) {
    std::vector<u32> ins;
    std::vector<size_t> loop_stack;
    for (unsigned char c : code) {
        if (c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f') continue;
        ins.push_back((u32)c);
        if (c == '[') { ins.push_back(0); loop_stack.push_back(ins.size() - 1); }
        else if (c == ']') {
            if (loop_stack.empty()) throw std::runtime_error("unbalanced ]");
            size_t start = loop_stack.back(); loop_stack.pop_back();
            ins[start] = (u32)ins.size();
            ins.push_back((u32)(start + 1));
        }
    }
    return ins;
}

struct Machine {
    std::vector<u32> code;
    std::vector<u32> ram;
    std::vector<u8> input; size_t in_pos = 0;
    std::vector<u8> output;
    Registers reg;
    std::vector<Registers> trace;
    static constexpr size_t DEFAULT_RAM_SIZE = 30000;  // machine.rs:114

    Machine(std::vector<u32> code_, std::vector<u8> input_, size_t ram_size = DEFAULT_RAM_SIZE)
        : code(std::move(code_)), ram(ram_size, 0), input(std::move(input_)) {}

    // machine.rs:141-161
    void execute() {
        while (reg.ip < code.size()) {
            reg.ci = code[reg.ip];
            reg.ni = (reg.ip == code.size() - 1) ? 0 : code[reg.ip + 1];
            trace.push_back(reg);
            step(reg.ci);
            reg.clk = (M31(reg.clk) + M31(1)).v;  // machine.rs:231-234
            reg.ip = (M31(reg.ip) + M31(1)).v;
        }
        reg.ci = 0; reg.ni = 0;
        trace.push_back(reg);
    }

   private:
    // machine.rs:177-229
    void step(u32 ins) {
        switch (ins) {
            case OP_RIGHT: reg.mp = (M31(reg.mp) + M31(1)).v; break;
            case OP_LEFT: reg.mp = (M31(reg.mp) - M31(1)).v; break;
            case OP_PLUS: ram.at(reg.mp) = (M31(ram.at(reg.mp)) + M31(1)).v; break;
            case OP_MINUS: ram.at(reg.mp) = (M31(ram.at(reg.mp)) - M31(1)).v; break;
            case OP_READCHAR:
                if (in_pos >= input.size()) throw std::runtime_error("input exhausted");
                ram.at(reg.mp) = input[in_pos++];
                break;
            case OP_PUTCHAR: output.push_back((u8)ram.at(reg.mp)); break;
            case OP_JZ: {
                u32 arg = code.at(reg.ip + 1);
                reg.ni = arg;
                if (ram.at(reg.mp) == 0) { reg.ip = arg; return; }
                reg.ip = (M31(reg.ip) + M31(1)).v;
                break;
            }
            case OP_JNZ: {
                u32 arg = code.at(reg.ip + 1);
                if (ram.at(reg.mp) != 0) { reg.ip = (M31(arg) - M31(1)).v; return; }
                reg.ip = (M31(reg.ip) + M31(1)).v;
                break;
            }
            default: throw std::runtime_error("invalid instruction");
        }
        reg.mv = ram.at(reg.mp);
        reg.mvi = reg.mv == 0 ? 0 : inv(M31(reg.mv)).v;
    }
};

}  // namespace orc
