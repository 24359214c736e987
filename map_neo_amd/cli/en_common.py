"""Shared by the drop-ins of the EN pipeline's exact-match stages (exact_document_deduplication,
paragraph, exact_substr_dedup): shard ingest into one packed corpus in HBM and gzip JSON-lines
output.  The corpus of one invocation is held in host memory once and in HBM once."""
from __future__ import annotations

import multiprocessing as mp
import os

import numpy as np

CONTENT_FIELD_NAME = "raw_content"          # EXD:4, PAR:12, SUB:17
DOC_ID_FIELD_NAME = "doc_id"                # SUB:18, PAR:35


def _parse(task):
    path, content_field, id_field = task
    from .. import hostio
    r = hostio.read_jsonl_shard(path, content_field, id_field, normalize_en=False, on_error="skip")
    return r["ids"], r["line_no"], r["cand"], r["text"], r["offsets"]


def load_corpus(input_dir: str, workers: int, device: int = 0, content_field: str = CONTENT_FIELD_NAME,
                id_field=DOC_ID_FIELD_NAME):
    """-> dict: files, eng, text/offs (device, packed, documents in file-then-line order),
    doc_file / doc_row (which file / physical line every document came from), doc_ids,
    rows_per_file [(ids, line_no, cand)]"""
    from .. import hostio
    files = hostio.list_files(input_dir)
    tasks = [(p, content_field, id_field) for p in files]
    import torch
    from ..minhash import Engine, padded_size
    if hostio._io() is not None:
        # the native reader holds no GIL (gunzip + scan are ~90 % of a shard's time): threads parse while this
        # thread brings the GPU up, and the parsed text is never pickled between processes
        from multiprocessing.pool import ThreadPool
        pool = ThreadPool(max(1, workers))
    else:
        pool = mp.get_context("fork").Pool(max(1, workers))       # forked before this process touches the GPU
    try:
        parsed = pool.imap(_parse, tasks)                         # in file order; every shard goes to HBM as it arrives
        eng = Engine(device)
        eng.warm_up()
        parts, offs, doc_file, doc_row, doc_ids, per_file, base = [], [np.zeros(1, np.uint64)], [], [], [], [], 0
        for fi, (ids, line_no, cand, text, o) in enumerate(parsed):
            if text.size:
                parts.append(eng.to_device(text))
            offs.append(o[1:] + np.uint64(base))
            base += int(text.size)
            doc_file.append(np.full(len(cand), fi, dtype=np.int64))
            doc_row.append(line_no[cand].astype(np.int64) if len(cand) else np.zeros(0, np.int64))
            doc_ids.extend(ids[i] for i in cand.tolist())
            per_file.append((ids, line_no, cand))
    finally:
        pool.close()
        pool.join()
    parts.append(torch.zeros(padded_size(base) - base, dtype=torch.uint8, device=eng.device))
    return {"files": files, "eng": eng, "text": torch.cat(parts), "offs": eng.to_device(np.concatenate(offs)),
            "doc_file": np.concatenate(doc_file) if doc_file else np.zeros(0, np.int64),
            "doc_row": np.concatenate(doc_row) if doc_row else np.zeros(0, np.int64), "doc_ids": doc_ids,
            "rows_per_file": per_file, "workers": workers}


def reorder_packed(eng, text, offs, perm):
    """Packed corpus with its documents permuted / selected (device): document k of the result = document perm[k].
    One pass over the selected text (mdd_gather_docs): no per-byte index tensors, so it scales with the corpus."""
    return eng.gather_docs(text, offs, perm)


def _write_part(task):
    out_path, ids, raw, offs = task
    from .. import hostio
    return hostio.write_jsonl_part(out_path, DOC_ID_FIELD_NAME, CONTENT_FIELD_NAME, ids, raw, offs)


def write_docs(output_dir: str, ids, text, offs, workers: int, docs_per_part: int = 0) -> int:
    """gzip JSON lines {"doc_id", "raw_content"} (the two columns PAR:27-28 / SUB:92 keep); the part size is chosen
    so that every worker gets several parts (gzip is the slow step), at most 100k documents each.  With
    libmdd_io.so the parts are written by threads straight from the one host copy of the text (the native writer
    runs without the GIL); without it by spawned processes running the equivalent Python loop."""
    from multiprocessing.pool import ThreadPool
    from .. import hostio
    os.makedirs(output_dir, exist_ok=True)
    if docs_per_part <= 0:
        docs_per_part = min(100_000, max(1000, -(-len(ids) // (4 * max(1, workers)))))
    raw, o = text.cpu().numpy(), offs.cpu().numpy().astype(np.uint64)
    native = hostio._io() is not None
    tasks = []
    for p, k0 in enumerate(range(0, len(ids), docs_per_part)):
        k1 = min(len(ids), k0 + docs_per_part)
        path = os.path.join(output_dir, "part-%05d.json.gz" % p)
        if native:
            tasks.append((path, ids[k0:k1], raw, o[k0:k1 + 1]))
        else:
            tasks.append((path, ids[k0:k1], raw[int(o[k0]):int(o[k1])].copy(), o[k0:k1 + 1] - o[k0]))
    if not tasks:
        return 0
    if workers <= 1 or len(tasks) == 1:
        return sum(map(_write_part, tasks))
    if native:
        with ThreadPool(min(workers, len(tasks))) as pool:
            return sum(pool.map(_write_part, tasks))
    with mp.get_context("spawn").Pool(min(workers, len(tasks))) as pool:   # the parent holds a GPU context: never fork it
        return sum(pool.map(_write_part, tasks))
