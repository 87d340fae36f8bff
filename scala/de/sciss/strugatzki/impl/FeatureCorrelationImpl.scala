/*
 *  FeatureCorrelationImpl.scala -- replacement of the reference's class of this name: the body hands the
 *  configuration to the GPU library and parses the `<match>` elements it returns.  The database is dealt over
 *  the GPUs named by -Dstrugatzki.devices (files in lexicographic order of their meta file names, round robin).
 */
package de.sciss.strugatzki
package impl

import de.sciss.processor.impl.ProcessorImpl

import scala.collection.breakOut
import scala.concurrent.blocking
import scala.xml.XML

private[strugatzki] final class FeatureCorrelationImpl(val config: FeatureCorrelation.Config)
  extends FeatureCorrelation with ProcessorImpl[FeatureCorrelation.Product, FeatureCorrelation] {

  import FeatureCorrelation._

  protected def body(): Product = blocking {
    Native.withContexts(Native.devices) { ctx =>
      val res = Native.run(this, p => { progress = p }, () => checkAborted()) { obs =>
        Native.runCorrelation(ctx, config.toXML.toString, obs)
      }
      (XML.loadString(res) \ "match").map(Match.fromXML)(breakOut): Product
    }
  }
}
