import sys, os, random, importlib
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("rust-compression_amd")
sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import oracle
eng = pkg.GpuEngine(0, 16)
rng = random.Random(3)
fib = [1, 1]
while len(fib) < 25:
    fib.append(fib[-1] + fib[-2])
tables = [fib[:20], fib[:25], [5] * 7, [1, 1, 1, 1, 2], [0] * 6, [3, 3, 2, 2, 1, 1, 1], [0, 0, 0]]
for _ in range(60):
    n = rng.randint(3, 258)
    r = rng.uniform(0.35, 0.9)
    f = [int(900000 * (1 - r) * r ** i * rng.uniform(0.7, 1.3)) for i in range(n)]
    rng.shuffle(f)
    tables.append(f)
for k in range(240):
    n = rng.choice([2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 33, 64, 100, 129, 200, 257, 258])
    mode = k % 5
    if mode == 0:
        f = [rng.randint(0, 3) for _ in range(n)]
    elif mode == 1:
        f = [rng.randint(0, 100000) for _ in range(n)]
    elif mode == 2:
        f = [1] * n
    elif mode == 3:
        f = [int(2 ** (rng.random() * 20)) for _ in range(n)]
    else:
        f = [rng.choice([0, 1, 5, 5, 5, 900]) for _ in range(n)]
    tables.append(f)
bad = 0
exps = [oracle.bzip2_code_lengths(f, 17) for f in tables]
for rep in range(1):
  for ti, f in enumerate(tables):
    exp, elm = exps[ti]
    res = [eng.debug_code_lengths(f) for _ in range(1)]
    for got, lm in res:
        if (got, lm) != (exp, elm):
            bad += 1
            idx = [i for i in range(len(f)) if got[i] != exp[i]]
            print("table", ti, "n", len(f), "lm", lm, elm, "bad idx", idx[:20], "got", [got[i] for i in idx[:20]], "exp", [exp[i] for i in idx[:20]])
print("bad", bad)
