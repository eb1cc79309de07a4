import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    m=d.get("moving") or {}
    print(f, "step %.3f syncs %.1f | 1m %.4f (syncs %.1f) | moving %.3f min %.3f max %.3f ahead %s missed %s" % (d["ms_per_step"], d["host_syncs_per_step"], (d.get("config_1m") or {}).get("ms_per_step", 0), (d.get("config_1m") or {}).get("host_syncs_per_step", 0), m.get("ms_per_step",0), m.get("ms_min",0), m.get("ms_max",0), m.get("sorts_ahead"), (m.get("plan_events") or {}).get("sorts_ahead_missed")))
