#!/bin/bash
SH="512,512,700,hinge 1024,512,700,hinge 384,1000,220,hinge 512,1000,220,hinge 1024,512,136,hinge 512,1000,136,dcg_hinge 2048,300,136,hinge 1024,400,64,hinge 300,512,700,hinge 512,1000,220,logistic 1024,512,136,logistic 2048,300,136,arp2 512,2000,64,hinge 256,4000,32,hinge"
echo "== parts (auto)"; python scripts/dev/parts_check.py --time --nocheck --shapes $SH 2>&1 | grep shape
echo "== old plans"; LTR_DISABLE_PARTS=1 python scripts/dev/parts_check.py --time --nocheck --shapes $SH 2>&1 | grep shape
