"""Scaled-down crossing of the code path tests/full_configs.py runs at the BASELINE sizes: block-streamed signatures
(dedup.signatures_streamed: text generated, hashed and dropped block by block) + ONE clustering call over more than
2^28 band records (a 2^30-slot bucket table, record indices beyond 2^28), with the size-independent property checks of
that script and an oracle sample.  30 M short documents (~200 B) keep it to some ten seconds."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_block_streamed_signatures_and_one_clustering_over_2_28_records(tmp_path):
    import json

    from map_neo_amd import corpus
    from map_neo_amd.minhash import Engine
    import full_configs as FC

    eng = Engine(0)
    n = (1 << 28) // 9 + 200_000          # 30.0 M documents -> 270 M band records
    spec = corpus.CorpusSpec(n, 200, seed=99, lang="mixed", dup_frac=0.15, edit_rates=FC.RATES, short_frac=0.001)
    sink = str(tmp_path / "full.jsonl")
    FC.run_config(eng, "scaled", spec, [(9, 13, 128)], 2_000_000, sink)
    res = json.loads(open(sink).read().splitlines()[-1])
    assert res["n_records"] > (1 << 28) and res["n_docs"] == n
    assert res["exact_copies"] > 100_000 and res["n_removed"] >= res["exact_copies"]
    assert res["copies_found_by_edit_rate"]["0"] == 1.0
    assert res["tag_collisions_36bit"] > res["tag_collisions_64bit"]
    eng.close()
