"""The Jacobi form of the single-kernel iteration forms r / c and q / c (c = 1 .. 6 neighbours) as  q = RN(a y), r = fma(-c, q, a),
q' = fma(r, y, q)  with y = RN(1 / c) from a table (oi_k_cg1b.hip: div_small) and claims the bits of the IEEE division (Markstein's
correction; |1/c - y| <= 2^-54 / c makes q faithful for these divisors).  The same three operations on the host, with hardware fma,
against `a / c` (the GPU side of the claim: test_single_kernel_iteration_two_kernels_are_bit_identical, whose other kernel divides)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_three_operation_quotient_is_the_correctly_rounded_one(tmp_path):
    exe = tmp_path / "div_small_check"
    subprocess.check_call(["gcc", "-O2", "-march=native", "-ffp-contract=off", "-o", str(exe), os.path.join(HERE, "csrc", "div_small_check.c"), "-lm"])
    out = subprocess.run([str(exe), "5000000"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    last = out.stdout.strip().splitlines()[-1]
    n, bad = int(last.split()[0]), int(last.split()[2])
    assert n >= 6 * 5000000 and bad == 0, out.stdout
