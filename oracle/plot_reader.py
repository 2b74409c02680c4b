"""What `ngs plot sample <JSON>` does with a `ngs qc` results file, restated for the tests (TEST INFRASTRUCTURE: only
tests/ may import this, like the rest of oracle/).  The reference's plots are the downstream consumer of the drop-in
boundary: a results file written by the GPU path has to be readable by them.

  Results::read                     src/qc/results.rs:63-67        serde_json into the typed structs: the shape check
                                                                    is tests/test_results_schema.py's `conforms`
  get_all_sample_plots              src/plot/command.rs:93-118     the three sample plots and the --only filter
  QualityScoreDistributionPlot      src/plot/sample/quality_score_distribution.rs:41-76
  GCContentDistributionPlot         src/plot/sample/gc_content_distribution.rs:37-77
  VariantAlleleFractionDistribution src/plot/sample/vaf_distribution.rs:37-73

Each function returns the numbers the plot draws (plotly itself is not restated) or raises PlotError with the
reference's message.  Quartiles come from the oracle's Histogram (oracle/histogram.c = utils/histogram.rs:272-352)."""
import ctypes as C
import os
from typing import Dict, List, Optional

from . import oracle_py

PLOTS = (("Quality Score Distribution", "quality-score-distribution"),      # command.rs:96-100 (this order)
         ("GC Content Distribution", "gc-content-distribution"),
         ("Variant Allele Fraction Distribution", "vaf-distribution"))


class PlotError(RuntimeError):
    pass


class _Hist:
    """A reference Histogram rebuilt from its serialized form {values, range_start, range_stop} (histogram.rs:152-159)."""

    def __init__(self, doc: dict):
        values, start, stop = doc["values"], doc["range_start"], doc["range_stop"]
        if start != 0 or len(values) != stop - start + 1:
            raise PlotError(f"histogram with range {start}..={stop} holds {len(values)} values")
        self.lib = oracle_py.load()
        self.h = oracle_py.Hist()
        if self.lib.orc_hist_init(C.byref(self.h), stop) != 0:
            raise MemoryError
        for b, v in enumerate(values):
            if v and self.lib.orc_hist_increment_by(C.byref(self.h), b, v) != 0:
                raise PlotError("bin out of bounds")
        self.values = list(values)

    def __del__(self):
        try:
            self.lib.orc_hist_free(C.byref(self.h))
        except Exception:
            pass

    def sum(self) -> int:
        return int(self.lib.orc_hist_sum(C.byref(self.h)))

    def quartile(self, name: str) -> Optional[float]:
        some, out = C.c_int(0), C.c_double(0.0)
        rc = getattr(self.lib, "orc_hist_" + name)(C.byref(self.h), C.byref(some), C.byref(out))
        if rc != 0:
            raise PlotError(f"{name}: the reference would panic here (no non-empty bin above the tie)")
        return out.value if some.value else None


def trace_name(path: str) -> str:
    return os.path.basename(path).replace(".results.json", "")     # e.g. gc_content_distribution.rs:80-86


def select_plots(only: Optional[str] = None) -> List[str]:
    """command.rs:93-118"""
    names = [n for n, _ in PLOTS]
    if only is not None:
        names = [n for n in names if n.lower() == only.lower()]     # eq_ignore_ascii_case
        if not names:
            raise PlotError(f"No plots matched the specified `--only` flag: {only}. Use `ngs list plots` to see the full "
                            "list of plots supported.")
    return names


def gc_content_distribution(results: dict, path: str) -> Dict[str, list]:
    gc = results.get("gc_content")
    if gc is None:
        raise PlotError(f"File {path} has no GC content information!")                    # :43-46
    hist = _Hist(gc["histogram"])
    total = hist.sum()
    if total == 0:
        raise PlotError(f"File {path} has GC content information, but it's empty!")       # :59-64
    return {"x": list(range(len(hist.values))), "y": [v / total for v in hist.values], "name": trace_name(path)}


def quality_score_distribution(results: dict, path: str) -> Dict[str, list]:
    qs = results.get("quality_scores")
    if qs is None:
        raise PlotError(f"File {path} has no quality score information!")                 # :45-51
    x, y, plus, minus, y_lim = [], [], [], [], 0.0
    for position in sorted(int(k) for k in qs["scores"]):                                 # scores.keys().sorted(): usize keys
        h = _Hist(qs["scores"][str(position)])
        median, q3, q1 = h.quartile("median"), h.quartile("third_quartile"), h.quartile("first_quartile")
        if median is None or q3 is None or q1 is None:
            raise PlotError(f"position {position}: median().unwrap() on an empty histogram")   # :63-65 would panic
        x.append(position)
        y.append(median)
        plus.append(q3 - median)
        minus.append(median - q1)
        y_lim = max(y_lim, q3)
    return {"x": x, "y": y, "error_plus": plus, "error_minus": minus, "y_lim": y_lim, "name": trace_name(path)}


def vaf_distribution(results: dict, path: str) -> Dict[str, list]:
    edits = results.get("edits")
    if edits is None:
        raise PlotError(f"File {path} has no Edits information!")                         # :42-45
    hist = _Hist(edits["vaf_histogram"])
    total = hist.sum()
    if total == 0:
        raise PlotError(f"File {path} has  information, but it's empty!")                 # :55-60 (sic)
    return {"x": list(range(len(hist.values))), "y": [v / total for v in hist.values], "name": trace_name(path)}


GENERATORS = {"Quality Score Distribution": quality_score_distribution, "GC Content Distribution": gc_content_distribution,
              "Variant Allele Fraction Distribution": vaf_distribution}


def plot_sample(results: dict, path: str, only: Optional[str] = None) -> Dict[str, dict]:
    """sample.rs:41-94: every selected plot's data, keyed by output file name (<filename>.sample.html)."""
    out = {}
    for name in select_plots(only):
        filename = dict(PLOTS)[name]
        out[filename + ".sample.html"] = GENERATORS[name](results, path)
    return out
