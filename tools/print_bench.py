import sys,json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    print(round(d["value"]), {k:round(v,3) for k,v in d["roofline"]["kernel_ms"].items()}, "serial", {k:round(v,3) for k,v in d["roofline"]["kernel_ms_serial"].items()}, "p50", round(d.get("p50_single_frame_us",0)), d["status_histogram"])
