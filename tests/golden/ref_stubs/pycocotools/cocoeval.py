class COCOeval: pass
