"""Structure-aware mutation of glTF documents: random JSON edits (indices, counts, types, cycles) on the tests' documents; the loader must load or
raise ValueError.  Children of 500 documents each: a crash or a hang shows as a dead child and the last document is kept.
    python3 tools/probes/fuzz_importers/make_seeds.py; python3 tools/probes/fuzz_importers/struct_fuzz.py <first seed> <documents>
On a sanitized build of the host library (round 6: 72 000 documents, clean):
    g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -shared -pthread -Irfw-rs_amd/host -o /tmp/fuzz/librfw_host_asan.so rfw-rs_amd/host/{rfw_host,gltf,gltf_export,jpeg,obj}.cpp -lz
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:allocator_may_return_null=1 RFW_HOST_ASAN=/tmp/fuzz/librfw_host_asan.so python3 ... struct_fuzz.py 0 8000"""
import sys, os, json, random, copy, pathlib, subprocess, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def walk(node, path, out):
    if isinstance(node, dict):
        for k, v in node.items(): walk(v, path + [k], out)
    elif isinstance(node, list):
        for i, v in enumerate(node): walk(v, path + [i], out)
    out.append(path)

def get(doc, path):
    for k in path: doc = doc[k]
    return doc
def setv(doc, path, v):
    for k in path[:-1]: doc = doc[k]
    doc[path[-1]] = v

VALUES = [0, 1, -1, 2, 3, 7, 255, 65535, 65536, 2**31 - 1, 2**31, 2**32 - 1, 2**40, -2**31, 1e30, -1e30, 0.5, "", "SCALAR", "VEC3", "VEC4", "MAT4", "LINEAR", "STEP", "CUBICSPLINE",
          "rotation", "translation", "scale", "weights", None, True, [], {}, [0], [0, 1, 2], 5120, 5121, 5122, 5123, 5125, 5126]

def mutate(doc, rng):
    paths = []
    walk(doc, [], paths)
    for _ in range(rng.randint(1, 4)):
        p = rng.choice(paths)
        if not p: continue
        try:
            cur = get(doc, p)
        except Exception:
            continue
        mode = rng.randint(0, 5)
        try:
            if mode <= 2:
                setv(doc, p, rng.choice(VALUES))
            elif mode == 3 and isinstance(cur, (int, float)) and not isinstance(cur, bool):
                setv(doc, p, cur + rng.choice([-1, 1, 1000, -1000, 2**31]))
            elif mode == 4:
                parent = get(doc, p[:-1])
                if isinstance(parent, dict): del parent[p[-1]]
                elif isinstance(parent, list): parent.pop(p[-1])
            else:
                parent = get(doc, p[:-1])
                if isinstance(parent, list): parent.append(copy.deepcopy(rng.choice(parent)) if parent else 0)
        except Exception:
            pass
    # node cycles, now and then
    if "nodes" in doc and isinstance(doc["nodes"], list) and doc["nodes"] and rng.random() < 0.15:
        n = len(doc["nodes"]); a, b = rng.randrange(n), rng.randrange(n)
        if isinstance(doc["nodes"][a], dict): doc["nodes"][a]["children"] = [b, a]

def child(seed_files, start, count, workdir):
    import rfw_rs_amd.scene as sc
    if os.environ.get('RFW_HOST_ASAN'): sc.HOST_LIB = os.environ['RFW_HOST_ASAN']
    from rfw_rs_amd import Scene
    rng = random.Random(start)
    ok = bad = 0
    for it in range(start, start + count):
        src = seed_files[it % len(seed_files)]
        doc = json.loads(open(src).read())
        mutate(doc, rng)
        d = pathlib.Path(workdir); shutil.rmtree(d, ignore_errors=True); shutil.copytree(os.path.dirname(src), d)
        p = d / os.path.basename(src)
        p.write_text(json.dumps(doc))
        open(workdir + ".last", "w").write(json.dumps(doc))
        try:
            s = Scene().load_gltf(str(p)); ok += 1
            try:
                s.set_animation_time(0.3); s.pose(0.7); s.set_animation_time(1.9)
            except (RuntimeError, ValueError):
                pass
        except ValueError:
            bad += 1
    print("ok", ok, "rejected", bad)

if __name__ == "__main__":
    if sys.argv[1] == "child":
        child(sys.argv[5:], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    else:
        seeds = ["/tmp/fuzz/seeds/ext/scene.gltf", "/tmp/fuzz/seeds/b64/scene.gltf", "/tmp/fuzz/seeds/tex/textured.gltf", "/tmp/fuzz/seeds/skin/skinned.gltf",
                 "/tmp/fuzz/seeds/anim/animated.gltf", "/tmp/fuzz/seeds/anim_jpeg/animated.gltf"]
        start, total, chunk = int(sys.argv[1]), int(sys.argv[2]), 500
        for s in range(start, start + total, chunk):
            w = "/tmp/fuzz/swork_%d" % start
            r = subprocess.run([sys.executable, __file__, "child", str(s), str(chunk), w] + seeds, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                print("CRASH at chunk", s, "rc", r.returncode, r.stderr[-400:])
                shutil.copy(w + ".last", "/tmp/fuzz/crash_%d.json" % s)
            else:
                print(s, r.stdout.strip())
