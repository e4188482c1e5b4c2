"""Single-frame (B = 1) drop-in latency through the compiled C++ consumer: 640x480 (configs[2] shape) and 1280x960
(configs[4]); prints one JSON object per run.  usage: python tools/measure_dropin.py [nframes]"""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dropin_harness as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
tmp = tempfile.mkdtemp()
for (w, h, nf, nl, style) in ((640, 480, 1000, 200, "struct"), (640, 480, 1000, 200, "desk"), (1280, 960, 2000, 200, "struct")):
    g, d = D.synth_stream(w, h, n if w == 640 else max(n // 2, 12), style, 20250418)
    path = os.path.join(tmp, f"frames_{w}_{style}.bin")
    D.write_frames(path, g, d)
    for stages in (False, True):
        r = D.run(path, nf, nl, 10 if w == 640 else 6, stages=stages)
        r["style"] = style
        r["stage_timers"] = stages
        print(json.dumps(r), flush=True)
