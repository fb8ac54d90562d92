"""Integer-exact host logic against the reference's outputs (tests/golden/augment_golden.json)."""
from pathlib import Path

from leaffliction_amd.dataio.manifest import ManifestItem, build_label_mapping
from leaffliction_amd.preprocessing.dataset_components import AugmentationPlanner
from leaffliction_amd.utils.confusion_matrix import compute_confusion_counts


def test_planner_matches_reference(golden):
    _, meta = golden
    plan = AugmentationPlanner(meta["planner"]["counts"]).calculate_plan()
    assert plan == meta["planner"]["plan"]
    assert [list(v) for v in plan.values()] == [list(v) for v in meta["planner"]["plan"].values()]


def test_label_mapping_matches_reference(golden):
    _, meta = golden
    items = [ManifestItem(id=str(i), plant="p", cls="c", label=lab, split="train", src=Path("x"))
             for i, lab in enumerate(meta["label_mapping"]["labels"])]
    assert build_label_mapping(items) == meta["label_mapping"]["label2idx"]


def test_confusion_counts_match_reference(golden):
    _, meta = golden
    c = meta["confusion"]
    assert compute_confusion_counts(c["y_true"], c["y_pred"], c["num_classes"]) == c["matrix"]
