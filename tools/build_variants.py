#!/usr/bin/env python3
"""Tuning aid: builds A/B variants of libmilzma.so from generator settings.

    tools/build_variants.py name[:ENV=VAL,ENV=VAL...][:+FLAG,+FLAG] ...

Each variant runs tools/gen_fast_loop.py with the given MILZMA_GEN_* environment, builds
lzma_rs_amd/variants/libmilzma_<name>.so (extra hipcc flags after '+'), and at the end the default
fast_loop_asm.inc is regenerated so that the tree stays in its committed configuration.
Run the variants with experiments/ab_bench.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lzma_rs_amd", "csrc")
BASE_FLAGS = ("-O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -structurizecfg-skip-uniform-regions=true "
              "-mllvm -simplifycfg-sink-common=false")


def gen(env):
    e = {k: v for k, v in os.environ.items() if not k.startswith("MILZMA_GEN_")}
    e.update(env)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_fast_loop.py")], env=e, stdout=subprocess.DEVNULL)


def main():
    os.makedirs(os.path.join(ROOT, "lzma_rs_amd", "variants"), exist_ok=True)
    try:
        for spec in sys.argv[1:]:
            parts = spec.split(":")
            name, env, flags = parts[0], {}, []
            for p in parts[1:]:
                for item in filter(None, p.split(",")):
                    if item.startswith("+"):
                        flags.append(item[1:])
                    else:
                        k, v = item.split("=", 1)
                        env[k if k.startswith("MILZMA_GEN_") else "MILZMA_GEN_" + k] = v
            gen(env)
            # the hazards inline asm must respect itself (v_readlane after a VALU write, VALU reads of fresh lane masks): linted per variant
            lint = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(ROOT, "tests", "test_host_abi.py") + "::test_asm_loop_wait_states"],
                                  capture_output=True, text=True, cwd=ROOT)
            if lint.returncode != 0:
                sys.stderr.write(lint.stdout[-1500:])
                raise SystemExit("variant %s fails the wait-state lint" % name)
            out = os.path.join("..", "variants", "libmilzma_%s.so" % name)
            r = subprocess.run(["make", "-C", CSRC, "-s", "-B", "CXXFLAGS=" + " ".join([BASE_FLAGS] + flags), "OUT=" + out],
                               capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise SystemExit("variant %s failed to build" % name)
            warn = [l for l in (r.stdout + r.stderr).splitlines() if "warning" in l]
            print("built %-24s env=%s flags=%s%s" % (name, env, flags, "  WARNINGS: %d" % len(warn) if warn else ""))
    finally:
        gen({})


if __name__ == "__main__":
    main()
