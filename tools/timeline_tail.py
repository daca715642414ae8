import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find last stats_kernel<false> occurrence (timed region's summary)
idx=[i for i,r in enumerate(rows) if 'stats_kernel<false>' in r['Kernel_Name']]
for which in idx[-2:]:
    base=int(rows[which-6]['Start_Timestamp'])
    print('----')
    for r in rows[which-6:which+8]:
        print(f"{(int(r['Start_Timestamp'])-base)/1e3:9.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}  q{r['Queue_Id']}  {r['Kernel_Name'][:70]}  grid {r['Grid_Size_X']}")
