"""Itemises the latency phase of every tree of a proof from a ZK_WG_TRACE build (ZK_BUILD_DEFS="-DZK_WG_TRACE=1"): per latency
launch the levels it took in each form of the hash, the time each level actually cost, and the shares of the launch boundary +
first load, the continuation hand-over and the PCIe post.  VERDICT r05 item 3(a): so that the gap between the measured
`merkle_top` and DESIGN.md 4.3's floor (levels x one hash latency) has names.

    ZK_BUILD_DEFS="-DZK_WG_TRACE=1" python -m zkstark_amd.build
    ZK_WG_TRACE_FILE=/tmp/wg.txt python tools/wg_trace.py run 21 sha256     # one traced proof at domain 2^(21+3)
    python tools/wg_trace.py report /tmp/wg.txt [--kernel-csv trace.csv]

The stamps are s_memrealtime ticks (100 MHz): 10 ns resolution, one time base for every compute unit.
"""
import sys

FORMS_SHA = (("one-lane", 256, 4.6), ("main/helper", 128, 4.9), ("quad", 64, 3.1))


def form_of(w, hash_kind):
    """The form merkle_wg_kernel uses for a level of w nodes per workgroup, and DESIGN.md's latency for one pass of it (us)."""
    if hash_kind == 0:
        if w <= 64:
            return "quad", 3.1
        if w <= 128:
            return "main/helper", 4.9
        return "one-lane", 4.6 * ((w + 255) // 256)
    if w <= 32:
        return "row16", 3.9 * ((w + 15) // 16)
    if w <= 64:
        return "quad", 6.0
    return "one-lane", 11.0 * ((w + 255) // 256)


def run(log_n, hash_name):
    sys.path.insert(0, '.')
    import zkstark_amd as zk
    with zk.Context(log_n, 3, hash=hash_name) as c:
        c.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        for _ in range(3):
            c.prove()
    # a fresh context for the traced proof: the dump happens when a context is destroyed and covers every launch since the last dump
    import os
    path = os.environ.get("ZK_WG_TRACE_FILE")
    if path and os.path.exists(path):
        os.unlink(path)
    with zk.Context(log_n, 3, hash=hash_name) as c:
        c.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        c.prove()
    print("traced one proof at domain 2^%d (%s)" % (log_n + 3, hash_name))


def parse(path):
    recs = []
    for line in open(path):
        if not line.startswith("wg "):
            continue
        head, stamps = line.split(":", 1)
        kv = dict(x.split("=") for x in head.split()[1:])
        st = [int(x) for x in stamps.split()]            # st[i - 1] = slot i, ticks relative to slot 1; -1 = not reached
        recs.append(({k: int(v) for k, v in kv.items()}, [None] + [(x * 0.01 if x >= 0 else None) for x in st]))
    return recs


def report(path):
    recs = parse(path)
    tot = {"launches": 0, "kernel_us": 0.0, "levels": 0, "level_us": 0.0, "model_us": 0.0, "load_us": 0.0, "cont_us": 0.0, "post_us": 0.0}
    byform = {}
    print("# per latency launch: depth_in -> depth_out, workgroups, [level: nodes per workgroup, form, measured us / modelled us] ...")
    for m, s in recs:
        hk, j, j2, depth = m["hash"], m["j"], m["j2"], m["depth_in"]
        cnt = 1 << j
        out = []
        load = s[2] if s[2] is not None else 0.0          # entry -> inputs hashed / loaded into LDS (slot 2 is re-stamped at level 1)
        lv_us = 0.0
        for phase, (base, levels, c0) in enumerate(((2, j, cnt), (24, j2, m["wgs"]))):
            if phase == 1 and not j2:
                break
            for t in range(1, levels + 1):
                a, b = s[base + t - 1], s[base + t]
                if a is None or b is None:
                    continue
                w = c0 >> t
                form, model = form_of(w, hk)
                d = b - a
                out.append(f"{w}:{form}:{d:.1f}/{model:.1f}")
                lv_us += d
                tot["levels"] += 1; tot["level_us"] += d; tot["model_us"] += model
                f = byform.setdefault((hk, form), [0, 0.0, 0.0])
                f[0] += 1; f[1] += d; f[2] += model
        cont = (s[16] - s[2 + j]) if (j2 and s[16] is not None and s[2 + j] is not None) else 0.0
        post = (s[42] - s[40]) if (s[42] is not None and s[40] is not None) else 0.0
        end = max(x for x in s[1:] if x is not None)
        # what lies between the end of the levels and the post's start is the wait for the slowest workgroup (the poster is the last one)
        print(f"  depth {depth:2d} -> {depth - j - j2:2d}  {'leaf ' if m['leaf'] else 'inner'} wgs {m['wgs']:4d}  stamps span {end:6.1f} us: load/leaf {load:5.1f}, levels {lv_us:5.1f}, "
              f"hand-over {cont:4.1f}, post {post:4.1f} | " + " ".join(out))
        tot["launches"] += 1; tot["kernel_us"] += end; tot["load_us"] += load; tot["cont_us"] += cont; tot["post_us"] += post
    print(f"# {tot['launches']} latency launches: stamped span {tot['kernel_us']:.1f} us = load / leaf hashing {tot['load_us']:.1f} + {tot['levels']} levels {tot['level_us']:.1f} "
          f"(model: {tot['model_us']:.1f}) + continuation hand-overs {tot['cont_us']:.1f} + PCIe posts {tot['post_us']:.1f} + waits for the slowest workgroup / launch tail "
          f"{tot['kernel_us'] - tot['load_us'] - tot['level_us'] - tot['cont_us'] - tot['post_us']:.1f}")
    for (hk, form), (n, d, mdl) in sorted(byform.items()):
        print(f"#   {'sha256' if hk == 0 else 'field '} {form:12s}: {n:3d} levels, measured {d / n:5.2f} us per level (model {mdl / n:5.2f})")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "sha256")
    else:
        report(sys.argv[2])
