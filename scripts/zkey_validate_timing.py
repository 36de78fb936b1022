import importlib, os, sys, time, tempfile
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
cg = importlib.import_module("collaborative-circom_amd")
import oracle_lib as orc
for lm in (18, 20):
    d = tempfile.mkdtemp(); zp, wp = os.path.join(d, "s.zkey"), os.path.join(d, "s.wtns")
    orc.make_synthetic(orc.BN254, lm, 5, zp, wp, threads=64)
    cg.host_zkey_validate(orc.BN254, zp)
    th, td = cg.host_zkey_validate(orc.BN254, zp)
    t0 = time.time(); z = orc.ZKey(orc.BN254, zp); t_cpu = time.time() - t0
    print(f"2^{lm}: zkey {os.path.getsize(zp)/1e6:.0f} MB: host read+decode {th:.3f} s, upload + GPU on-curve/subgroup validation of 5 x 2^{lm} points {td:.3f} s; oracle CPU reader (no point checks) {t_cpu:.3f} s")
