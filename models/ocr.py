"""``models.ocr`` — imported by the reference's test_sr.py:6 (``from models import networks, ocr``) but never used by it: the
script recognises characters with the modelscope OCR pipeline (test_sr.py:55, utils/yolo_ocr_xloc.py), not with this legacy
transformer-OCR file.  Kept as an empty module so that the script's import line resolves against this package; the OCR /
detector front-end itself is outside the hot path (SURVEY.md §8f NEXT-4)."""
