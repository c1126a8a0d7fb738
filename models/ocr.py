"""``models.ocr`` — imported by the reference's test_sr.py:6 (``from models import networks, ocr``) but never used by it: the
script recognises characters with the modelscope OCR pipeline (test_sr.py:55, utils/yolo_ocr_xloc.py), not with this legacy
transformer-OCR file.  Kept as an empty module so that the script's import line resolves against this package; the OCR /
detector front-end itself is outside the hot path (SURVEY.md §8f NEXT-4)."""


def __getattr__(name):
    raise AttributeError(
        "models.ocr.%s: the reference's legacy models/ocr.py (transformer OCR; dead code for test_sr.py / test_w.py, which recognise characters with "
        "the modelscope pipeline, test_sr.py:55) is not part of this build — this module exists only so that `from models import networks, ocr` "
        "(test_sr.py:6) resolves.  Labels come from the caller, or from the encoder itself: MarconetPipeline.forward_blind." % name)
