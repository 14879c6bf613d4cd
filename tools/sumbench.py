import sys,json
for f in sys.argv[1:]:
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-500:]); continue
    print(f, round(j["value"]), round(j["ms_per_step"],2))
    for n,v in list(j["roofline"]["kernels"].items())[:9]: print("   ", n, v["launches_per_step"], v["avg_us"], v["ms_per_step"], v["tflops"])
