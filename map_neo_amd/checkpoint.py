"""Checkpoint manifest of the dedup stage ("checkpoint / resume").

The reference's stage boundaries are its checkpoints: EN persists the signatures (EN:128-129, never reloaded), ZH
writes every intermediate to disk (band pickles ZH-1:58-61, pair files ZH-2:35, component pickle ZH-3:58-59).  Here the
same boundaries are plain .npy / .json files named by one manifest.json, and a run can start from any of them:

    signatures   uint32 [n, K]     signatures.npy   (rows = candidate documents in input order)  + doc_ids.json
    edges        uint32 [m, 2]     edges.npy        (document index pairs: the bucket collisions)
    labels       uint32 [n]        labels.npy       (smallest document index of the component)

Band records are not persisted: they are a 0.3 ms function of the signatures.  The manifest records the parameters
that produced the artifacts; resuming with different parameters is refused.
"""
from __future__ import annotations

import json
import os
from typing import Optional

import numpy as np

STAGES = ("signatures", "edges", "labels")
FILES = {"signatures": "signatures.npy", "doc_ids": "doc_ids.json", "edges": "edges.npy", "labels": "labels.npy"}
MANIFEST = "manifest.json"


class CheckpointError(RuntimeError):
    pass


# parameters that shape the SIGNATURES (everything else only shapes edges / labels)
SIG_KEYS = ("width", "ngram_unit", "num_perm", "seed", "normalize", "id_field", "content_field")


def params_of(args) -> dict:
    """What produced the artifacts: the LSH parameters and what was read from the input (which field, which id, whether
    EN:31-40 was applied).  Where normalisation ran (--normalize_on) is not a parameter: host and device agree."""
    p = {"width": int(args.width), "ngram_unit": str(args.ngram_unit), "num_perm": int(args.num_perm), "bands": int(args.bands),
         "rows": int(args.rows), "seed": int(args.seed),
         **({"min_jaccard": float(args.min_jaccard)} if getattr(args, "min_jaccard", None) is not None else {})}
    if hasattr(args, "content_field_name"):          # the EN command: always normalises, ids are doc_id (EN:123)
        p.update(normalize="en", id_field="doc_id", content_field=str(args.content_field_name))
    else:
        p.update(normalize=str(getattr(args, "normalize", "none")), id_field=str(getattr(args, "id_field", "")),
                 content_field=str(getattr(args, "content_field", "")))
    return p


def ids_digest(doc_ids) -> str:
    import hashlib
    h = hashlib.sha256()
    for d in doc_ids:
        h.update(json.dumps(d, ensure_ascii=False).encode("utf8", "surrogatepass"))
        h.update(b"\n")
    return h.hexdigest()


def _atomic_save(path: str, arr: np.ndarray) -> None:
    tmp = path + ".tmp.npy"
    np.save(tmp, arr)
    os.replace(tmp, path)


def save(directory: str, params: dict, doc_ids, signatures: Optional[np.ndarray] = None, edges: Optional[np.ndarray] = None,
         labels: Optional[np.ndarray] = None, extra: Optional[dict] = None) -> dict:
    """Write the given artifacts and (last, atomically) the manifest that names them.  Artifacts of an existing
    manifest with the same parameters AND the same documents (digest of the id list) that are not given again stay
    listed (a run resumed from `edges` does not hold the signatures, but they are still there) -- except that fresh
    signatures retire the edges / labels derived from older ones: a new run in a reused directory never lists a stale
    artifact next to a new one, whenever it stops."""
    os.makedirs(directory, exist_ok=True)
    arts = {}
    n = len(doc_ids)
    digest = ids_digest(doc_ids)
    try:
        with open(os.path.join(directory, MANIFEST), encoding="utf8") as f:
            old = json.load(f)
        if (old.get("version") == 1 and old.get("params") == params and old.get("n_candidates") == n
                and old.get("doc_ids_sha256") == digest):
            arts = {k: v for k, v in old["artifacts"].items() if os.path.exists(os.path.join(directory, v["file"]))}
    except (OSError, ValueError, KeyError):
        pass
    if signatures is not None:
        arts.pop("edges", None)
        arts.pop("labels", None)
    tmp = os.path.join(directory, FILES["doc_ids"] + ".tmp")
    with open(tmp, "w", encoding="utf8") as f:
        json.dump(doc_ids, f, ensure_ascii=False)
    os.replace(tmp, os.path.join(directory, FILES["doc_ids"]))
    arts["doc_ids"] = {"file": FILES["doc_ids"], "count": n}
    if signatures is not None:
        assert signatures.shape == (n, params["num_perm"]) and signatures.dtype == np.uint32
        _atomic_save(os.path.join(directory, FILES["signatures"]), signatures)
        arts["signatures"] = {"file": FILES["signatures"], "shape": [n, params["num_perm"]], "dtype": "uint32"}
    if edges is not None:
        assert edges.ndim == 2 and edges.shape[1] == 2 and edges.dtype == np.uint32
        _atomic_save(os.path.join(directory, FILES["edges"]), edges)
        arts["edges"] = {"file": FILES["edges"], "shape": list(edges.shape), "dtype": "uint32"}
    if labels is not None:
        assert labels.shape == (n,) and labels.dtype == np.uint32
        _atomic_save(os.path.join(directory, FILES["labels"]), labels)
        arts["labels"] = {"file": FILES["labels"], "shape": [n], "dtype": "uint32"}
    man = {"format": "map_neo_amd dedup checkpoint", "version": 1, "params": params, "n_candidates": n,
           "doc_ids_sha256": digest, "artifacts": arts, "extra": extra or {}}
    tmp = os.path.join(directory, MANIFEST + ".tmp")
    with open(tmp, "w", encoding="utf8") as f:
        json.dump(man, f, indent=1)
    os.replace(tmp, os.path.join(directory, MANIFEST))
    return man


def load(directory: str, stage: str, params: dict, input_dir: Optional[str] = None) -> dict:
    """Artifacts needed to resume from `stage`: {"manifest", "doc_ids", stage: ndarray}.  Refuses a manifest whose
    parameters differ from `params` (for `signatures` those that shape signatures, SIG_KEYS; for `labels` / `edges` all
    of them), whose recorded input directory differs from `input_dir` (when both are known), or whose id list does
    not match its recorded digest."""
    if stage not in STAGES:
        raise CheckpointError(f"--resume_from must be one of {STAGES}, not {stage!r}")
    mpath = os.path.join(directory, MANIFEST)
    if not os.path.exists(mpath):
        raise CheckpointError(f"{mpath} not found: nothing to resume from")
    with open(mpath, encoding="utf8") as f:
        man = json.load(f)
    if man.get("version") != 1 or "artifacts" not in man:
        raise CheckpointError(f"{mpath}: unknown manifest format")
    keys = SIG_KEYS if stage == "signatures" else sorted(set(params) | set(man["params"]))
    diff = {k: (man["params"].get(k), params.get(k)) for k in keys if man["params"].get(k) != params.get(k)}
    if diff:
        raise CheckpointError(f"{mpath} was written with other parameters (manifest, now): {diff}")
    was = man.get("extra", {}).get("input_dir")
    if input_dir is not None and was is not None and os.path.abspath(input_dir) != was:
        raise CheckpointError(f"{mpath} was written for input {was!r}, not {os.path.abspath(input_dir)!r}")
    if stage not in man["artifacts"]:
        raise CheckpointError(f"{mpath} lists no {stage!r} artifact")
    with open(os.path.join(directory, man["artifacts"]["doc_ids"]["file"]), encoding="utf8") as f:
        doc_ids = json.load(f)
    arr = np.load(os.path.join(directory, man["artifacts"][stage]["file"]), allow_pickle=False)
    if list(arr.shape) != man["artifacts"][stage]["shape"] or arr.dtype != np.uint32 or len(doc_ids) != man["n_candidates"]:
        raise CheckpointError(f"{directory}: {stage} artifact does not match its manifest entry")
    if man.get("doc_ids_sha256") not in (None, ids_digest(doc_ids)):
        raise CheckpointError(f"{directory}: doc_ids.json does not match the digest in the manifest")
    return {"manifest": man, "doc_ids": doc_ids, stage: arr}
