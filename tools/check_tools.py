"""Do the scripts in tools/ still run against the library as it is (C ABI version, pit_hip._lib, the switches of pit_hip.modules.unet)?
tools/ = this round's recipes; tools/convstack/ = the (frozen) conv stack's benches and diagnoses; tools/calibration/ = hardware probes.
Static checks only (no GPU needed): every .py byte-compiles, every attribute it reads from `_lib` / `U` (= pit_hip.modules.unet) /
`bench` exists today, every .sh parses; `--hip`: every .hip probe compiles for gfx950.  Exit code 1 lists what is stale.

    python tools/check_tools.py [--hip]
"""
import ast
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))


def attr_uses(tree, names):
    """{alias: {attribute, ...}} for `alias.attribute` reads where alias is one of `names`."""
    out = {n: set() for n in names}
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in out:
            out[node.value.id].add(node.attr)
    return out


def main():
    from pit_hip import _lib
    from pit_hip.modules import unet

    import bench

    targets = {"_lib": _lib, "U": unet, "unet": unet, "bench": bench}
    bad = []
    files = sorted(os.path.relpath(os.path.join(d, f), HERE) for d, _, fs in os.walk(HERE) for f in fs if "__pycache__" not in d)
    for f in files:
        path = os.path.join(HERE, f)
        if f.endswith(".py"):
            try:
                tree = ast.parse(open(path).read(), filename=f)
            except SyntaxError as e:
                bad.append(f"{f}: does not compile: {e}")
                continue
            imported = set()
            for node in ast.walk(tree):          # only aliases the script really binds to these modules
                if isinstance(node, ast.ImportFrom) and node.module:
                    for a in node.names:
                        name = a.asname or a.name
                        if (node.module.startswith("pit_hip") and a.name in ("_lib", "unet")) or (a.name == "unet" and name == "U"):
                            imported.add(name)
                if isinstance(node, ast.Import):
                    for a in node.names:
                        if a.name == "bench":
                            imported.add(a.asname or "bench")
            for alias, attrs in attr_uses(tree, imported & set(targets)).items():
                for a in sorted(attrs):
                    if not hasattr(targets[alias], a):
                        bad.append(f"{f}: {alias}.{a} no longer exists")
        elif f.endswith(".sh"):
            r = subprocess.run(["bash", "-n", path], capture_output=True, text=True)
            if r.returncode:
                bad.append(f"{f}: bash -n: {r.stderr.strip()}")
            for line in open(path):              # scripts it calls must exist
                for tok in line.replace('"', " ").replace("'", " ").split():
                    if "tools/" in tok and tok.endswith((".py", ".sh")):
                        ref = tok.split("tools/")[-1]
                        if not os.path.exists(os.path.join(HERE, ref)):
                            bad.append(f"{f}: calls tools/{ref}, which does not exist")
        elif f.endswith(".hip") and "--hip" in sys.argv:
            r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-c", "-o", os.devnull, path], capture_output=True, text=True)
            if r.returncode:
                bad.append(f"{f}: hipcc: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else 'failed'}")
    print(f"checked {len(files)} files in tools/ against ABI {_lib.ABI_VERSION}: {len(bad)} stale reference(s)")
    for b in bad:
        print("  STALE", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
