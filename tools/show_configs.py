"""tools/show_configs.py <configs.json> — readable dump of tools/config_runs.py's output."""
import json, sys
for c in json.load(open(sys.argv[1])):
    print({k: v for k, v in c.items() if k not in ("phases_readme_ms", "config")})
    for l, m in (c.get("phases_readme_ms") or []):
        print("      %8.3f  %s" % (m, l))
