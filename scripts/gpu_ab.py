"""Developer A/B on one box: runs bench.py (no CPU baseline) once per library variant built by scripts/ab_build.py and
prints one compact line each.   python scripts/gpu_ab.py [--extras] name1 name2 ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extras = "--extras" in args
names = [a for a in args if not a.startswith("--")]
for name in names:
    env = dict(os.environ)
    if name != "default":
        env["GPHIP_LIB"] = os.path.join(ROOT, "bayesianinference_amd", "lib", "variants", f"libgphip_{name}.so")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline"]
    if not extras:
        cmd.append("--no-extras")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    try:
        b = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        print(name, "FAILED", r.stderr[-400:])
        continue
    line = (f"{name:10s} ms/step {b['ms_per_step']:7.2f}  syrk in-run {b['roofline']['frac']:.3f}  alone "
            f"{b.get('roofline_syrk_alone', {}).get('frac', 0):.3f}  kbuild {b['roofline_kbuild']['frac']:.3f} "
            f"({b['roofline_kbuild']['avg_launch_ms']:.3f} ms)")
    oc = b.get("other_configs", {})
    if oc:
        c5 = oc.get("cfg5_matern52_n65536_d16_f32", {})
        line += (f"  cfg2 {oc.get('cfg2_n8192_d8_f64', {}).get('ms_per_eval', 0):.2f} ms  cfg4 "
                 f"{oc.get('cfg4_batch_200x4096_f64', {}).get('tflops', 0):.1f} TF  cfg5 fit {c5.get('fit_ms', 0):.0f} ms "
                 f"pred {c5.get('predict_10k_ms', 0):.0f} ms kb32 {c5.get('roofline_kbuild_f32', {}).get('frac', 0):.3f}")
    print(line, flush=True)
