#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
D="'down3.kernel','down2.kernel','down1.kernel','down0.kernel'"
bash tools/ab.sh $O/ab40.txt "" "--step-attr wgrad_alt=($D,'up2.kernel','up0.kernel','down6.kernel','down4.kernel')" "--step-attr wgrad_alt=($D,'up3.kernel','up1.kernel','down7.kernel','down5.kernel')" "--step-attr wgrad_alt=($D,'up2.kernel','down5.kernel')" "--step-attr wgrad_alt=('up2.kernel','up0.kernel','down6.kernel','down4.kernel','down2.kernel','down0.kernel')"
