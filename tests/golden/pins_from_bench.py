#!/usr/bin/env python3
"""The "c5" entry of c4_stream_pins.json (BASELINE configs[4] at full size, -l 1, 24 chunks) from a bench.py line whose CPU leg ran in mode
`whole_chunks`: there the REFERENCE (oracle/_ref/shmr_index over all 24 chunks, then oracle/_ref/shmr_overlap -t 24 -c c -M 240 for 8 whole chunks, on
the GPU box's host cores, from the seqdb files written from the device-resident read set) hashed its own streams -- `cpu_baseline.reference_streams`:
record count + SHA-256 with the padding bytes 27 and 60..63 of every record zeroed (formats.masked_stream_sha256) -- exactly what
make_c4_stream_pins.py pins for configs[3].  The reference leg costs ~16 minutes of 8 + 24 host processes; running it once inside the bench line
(which also compares the streams with the GPU's, `records_match_gpu`) instead of twice is why this entry is taken from the line.

  python tests/golden/pins_from_bench.py profiles/r06_bench_c5.json        # adds / replaces the workload's entry
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PINS = os.path.join(ROOT, "tests", "golden", "c4_stream_pins.json")


def main():
    src = sys.argv[1]
    d = json.loads(open(src).read().strip().splitlines()[-1])
    cb = d["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["mode"] == "whole_chunks" and cb["reference_streams"], "not a whole_chunks reference leg"
    wl = d["config"]["workload"].split(":")[0]
    T = cb["chunking"]
    entry = {"workload": wl, "genome_mb": None, "chunks": T, "levels": 1 if " l=1," in d["config"]["workload"] else 2, "mc_upper": 240,
             "reads": d["config"]["reads"], "seqdb_bytes": d["config"]["bases"], "read_set_hash": d["read_set_hash"], "seqdb_sha256": d.get("seqdb_sha256"),
             "streams": [{"chunk": r["chunk"], "records": r["records"], "masked_sha256": r["masked_sha256"]} for r in cb["reference_streams"]],
             "reference_overlap_leg_s": cb["overlap_s"], "reference_overlap_processes": cb["overlap_processes"], "reference_index_leg_s": cb["index_s"],
             "reference_index_processes": cb["index_processes"], "host_mem_total_gb": cb.get("host_ram_gb"),
             "made_by": "tests/golden/pins_from_bench.py %s: the reference leg of `python bench.py --workload %s --cpu-baseline whole_chunks` on the GPU box "
                        "(oracle/_ref/shmr_index -t %d -l %d over all chunks, oracle/_ref/shmr_overlap -t %d -c c -M 240 for the listed chunks); masked = bytes 27 "
                        "and 60..63 of every record zeroed" % (os.path.relpath(os.path.abspath(src), ROOT), wl, T, 1 if " l=1," in d["config"]["workload"] else 2, T)}
    pins = json.load(open(PINS))
    pins[wl] = entry
    json.dump(pins, open(PINS, "w"), indent=1)
    print("pinned %s: %d of %d chunks, %d records" % (wl, len(entry["streams"]), T, sum(s["records"] for s in entry["streams"])))


if __name__ == "__main__":
    main()
