"""Annotation writers: product `annotations.py` and the oracle restatement
`oracle/annot_ref.py` against golden strings produced by the reference's OWN TextBlock.to_dict /
xyxy2yolo / get_yololabel_strings (oracle/gen_golden_annot.py).  CPU only."""
import copy
import json
import os

import numpy as np
import pytest

from conftest import pkg
from oracle import annot_ref as A
from oracle import postproc_ref as R
from test_post_host import fake_outputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def blocks(seed, which):
    page, mask_u8, prob, blks = fake_outputs(seed, 512)
    H, W = prob.shape
    boxes, scores = R.boxes_from_bitmap(prob, prob > 0.3, W, H)
    lines = boxes[scores > 0.6].astype(np.int32)
    if which == "oracle":
        return R.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask_u8), W, H, page, mask_u8
    return pkg().textblock.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask_u8), W, H, page, mask_u8


@pytest.mark.parametrize("seed", [0, 1])
def test_text_formats_match_reference_golden(seed):
    gold = json.load(open(os.path.join(GOLD, f"annot_seed{seed}.json"), encoding="utf8"))
    p = pkg()
    # oracle restatement == the reference's own functions
    oblks, W, H, _, _ = blocks(seed, "oracle")
    texts = A.page_annotation_texts("page.jpg", W, H, oblks, save_dir="out", save_json=True)
    assert texts[os.path.join("out", "page.txt")] == gold["yolo"]
    assert texts[os.path.join("out", "line-page.txt")] == gold["lines"]
    assert texts[os.path.join("out", "page.json")] == gold["json"]
    assert list(oblks[0].to_dict().keys()) == gold["record_keys"]
    # product == golden, byte for byte (schema order, int/float types, number formatting)
    blks, W, H, _, _ = blocks(seed, "product")
    assert len(blks) == gold["n_blocks"]
    assert p.annotations.yolo_labels(blks, W, H) == gold["yolo"]
    assert p.annotations.line_polys_text(blks) == gold["lines"]
    assert p.annotations.blocks_json(blks) == gold["json"]
    assert list(blks[0].to_dict().keys()) == gold["record_keys"]


def test_page_files_names_png_roundtrip_and_empty_page(tmp_path):
    from PIL import Image
    p = pkg()
    ann = p.annotations
    blks, W, H, page, mask = blocks(0, "product")
    files = ann.page_files(str(tmp_path), "Scan.Page-01.JPG", page, mask, blks, save_json=True)
    names = sorted(os.path.basename(k) for k in files)
    assert names == ["Scan.Page-01.json", "Scan.Page-01.png", "Scan.Page-01.txt", "line-Scan.Page-01.txt",
                     "mask-Scan.Page-01.png"]
    assert A.png_name("dir/Scan.Page-01.JPG") == ann.png_path("dir/Scan.Page-01.JPG") == "dir/Scan.Page-01.png"
    assert A.png_name("dir/noext") == ann.png_path("dir/noext") == "dir/noext.png"
    ann.write_files(files)
    back = np.asarray(Image.open(tmp_path / "Scan.Page-01.png").convert("RGB"))[:, :, ::-1]
    np.testing.assert_array_equal(back, page)                       # PNG is lossless: BGR pixels survive
    np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "mask-Scan.Page-01.png")), mask)
    np.testing.assert_array_equal(ann.imread(str(tmp_path / "Scan.Page-01.png")), page)
    # a page without blocks: empty label file, no line file, "[]" json (reference inference.py:46-47,59)
    empty = ann.page_files(str(tmp_path), "e.png", page, mask, [], save_json=True)
    assert empty[str(tmp_path / "e.txt")] == "" and empty[str(tmp_path / "e.json")] == "[]"
    assert not any("line-" in k for k in empty)
    assert A.page_annotation_texts("e.png", W, H, [], save_dir=str(tmp_path)) == \
        {k: v for k, v in empty.items() if isinstance(v, str)}


def test_find_all_imgs_and_imread_variants(tmp_path):
    from PIL import Image
    ann = pkg().annotations
    rgb = np.random.RandomState(0).randint(0, 256, (20, 30, 3)).astype(np.uint8)
    Image.fromarray(rgb).save(tmp_path / "a.PNG")
    Image.fromarray(rgb).save(tmp_path / "b.bmp")
    Image.fromarray(rgb[:, :, 0]).save(tmp_path / "grey.png")
    Image.fromarray(np.dstack([rgb, np.full((20, 30), 128, np.uint8)])).save(tmp_path / "alpha.png")
    (tmp_path / "notes.txt").write_text("x")
    (tmp_path / "c.webp").write_bytes(b"")
    got = sorted(ann.find_all_imgs(str(tmp_path)))
    assert got == ["a.PNG", "alpha.png", "b.bmp", "grey.png"]
    assert all(os.path.isabs(q) for q in ann.find_all_imgs(str(tmp_path), abs_path=True))
    np.testing.assert_array_equal(ann.imread(str(tmp_path / "a.PNG")), rgb[:, :, ::-1])
    g = ann.imread(str(tmp_path / "grey.png"))
    assert g.shape == (20, 30, 3) and (g[..., 0] == g[..., 1]).all()            # grey -> 3 equal channels
    np.testing.assert_array_equal(ann.imread(str(tmp_path / "alpha.png")), rgb[:, :, ::-1])   # alpha dropped


def test_record_encoder_types():
    ann = pkg().annotations
    rec = {"a": np.int16(3), "b": np.float32(0.5), "c": np.bool_(True), "d": np.arange(3), "e": np.float64(1.25), "f": None}
    assert json.dumps(rec, cls=ann.RecordEncoder) == json.dumps(rec, cls=A.NumpyEncoder) == \
        '{"a": 3, "b": 0.5, "c": true, "d": [0, 1, 2], "e": 1.25, "f": null}'
