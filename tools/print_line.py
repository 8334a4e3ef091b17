"""Print the headline fields of a bench.py JSON line (a file holding the line, or stdin)."""
import json
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
line = json.loads([t for t in txt.strip().splitlines() if t.startswith("{")][-1])
r = line["roofline"]
print(f"{line['config']['workload'][:28]}: {line['value']} {line['unit']}, step {line['ms_per_step']} ms, stages {line['stages_ms']['encoder']} / "
      f"{line['stages_ms']['quantiser']} / {line['stages_ms']['decoder']} ms; kernel {r['kernel'][:24]} {r.get('avg_launch_us')} us, frac {r.get('frac')}; "
      f"whole call b2b {r.get('whole_call', {}).get('back_to_back_us')} us; visited {r.get('visited')}; "
      f"reference leg {line.get('reference_gpu_path', {}).get('images_per_s')} images/s (x{line.get('reference_gpu_path', {}).get('product_over_reference')})")
