class COCO: pass
