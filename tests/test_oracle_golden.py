"""The CPU oracle (oracle/align_oracle.c) against the golden vectors generated from
the reference itself (tests/golden/make_golden.py).  This is what pins the oracle on
a machine where /root/reference does not exist."""
from .conftest import load_golden, tup


def test_kats_locate(oracle):
    kats = load_golden("kats.json")
    assert len(kats["locate"]) > 200
    for c in kats["locate"]:
        got = oracle.locate(c["ref"], c["query"], c["e"], c["flags"], c["wr"], c["wq"], c["mo"], c["ic"])
        assert got == tup(c["out"]), c


def test_kats_compare(oracle):
    kats = load_golden("kats.json")
    for c in kats["compare_prefixes"]:
        assert oracle.compare_prefixes(c["ref"], c["query"], c["wr"], c["wq"]) == tup(c["out"]), c
    for c in kats["compare_suffixes"]:
        assert oracle.compare_suffixes(c["ref"], c["query"], c["wr"], c["wq"]) == tup(c["out"]), c


def test_kats_multi(oracle):
    for c in load_golden("kats.json")["multi_locate"]:
        got = oracle.multi_locate(c["ref"], c["query"], c["e"], c["flags"], c["mo"], c["mx"])
        assert got == [tup(x) for x in c["out"]], c


def _insert_norm(res):
    if res is None:
        return None
    return [list(res[0]), None if res[1] is None else list(res[1]), None if res[2] is None else list(res[2])]


def test_kats_insert(oracle):
    for c in load_golden("kats.json")["match_insert"]:
        orc = oracle.InsertOracle(c["a1"], c["a2"], **c["kw"])
        assert _insert_norm(orc.match_insert(c["r1"], c["r2"])) == c["out"], c


def test_locate_fuzz(oracle):
    cases = load_golden("locate_fuzz.json.gz")
    assert len(cases) == 6000
    for c in cases:
        got = oracle.locate(c["ref"], c["query"], c["e"], c["flags"], c["wr"], c["wq"], c["mo"], c["ic"])
        assert got == tup(c["out"]), c


def test_multi_fuzz(oracle):
    for c in load_golden("multi_fuzz.json.gz"):
        got = oracle.multi_locate(c["ref"], c["query"], c["e"], c["flags"], c["mo"], c["mx"])
        exp = None if c["out"] is None else [tup(x) for x in c["out"]]
        assert got == exp, c


def test_prefix_fuzz(oracle):
    for c in load_golden("prefix_fuzz.json.gz"):
        assert oracle.compare_prefixes(c["ref"], c["query"], c["wr"], c["wq"]) == tup(c["prefix"]), c
        assert oracle.compare_suffixes(c["ref"], c["query"], c["wr"], c["wq"]) == tup(c["suffix"]), c


def test_insert_fuzz(oracle):
    cache = {}
    for c in load_golden("insert_fuzz.json.gz"):
        key = (c["a1"], c["a2"], repr(sorted(c["kw"].items())))
        if key not in cache:
            cache[key] = oracle.InsertOracle(c["a1"], c["a2"], **c["kw"])
        assert _insert_norm(cache[key].match_insert(c["r1"], c["r2"])) == c["out"], c


def test_rmp_values(oracle):
    for k, size, p, q, rep in load_golden("rmp.json"):
        assert repr(oracle.rmp(k, size, p, q)) == rep, (k, size, p, q)


def test_synth_heads(oracle):
    """The synthetic workload generator reproduces the reads the golden outputs were
    computed on, and the oracle reproduces the reference's outputs on them."""
    from atropos_amd import synth
    heads = load_golden("synth_heads.json.gz")

    def rows(t):
        return [bytes(x.tolist()).decode("ascii") for x in t]

    for name in ("C1", "C2"):
        w = synth.workload(name, 0, heads[name]["count"])
        outs = [oracle.locate(w["adapter"], q, w["max_error_rate"], 14, False, False, w["min_overlap"],
                              w["indel_cost"]) for q in rows(w["reads"])]
        assert outs == [tup(x) for x in heads[name]["out"]]
        assert sum(o is not None for o in outs) > heads[name]["count"] // 3
    for name in ("C3", "C5"):
        w = synth.workload(name, 0, heads[name]["count"])
        orc = oracle.InsertOracle(w["adapter1"], w["adapter2"], **heads[name]["kw"])
        outs = [_insert_norm(orc.match_insert(a, b)) for a, b in zip(rows(w["reads1"]), rows(w["reads2"]))]
        assert outs == heads[name]["out"]
